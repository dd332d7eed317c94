#!/bin/bash
# round 5: fuzz of the changed backward / loss kernels, PMC passes over the training step with the final library, full GPU suite
set -o pipefail
mkdir -p gpurun_out
{
echo "# Round 5 fuzz campaign (update kernel: pair loads up front, wave-level bucket scan, early parameter-row prefetch), one gpurun call"
echo "## fuzz_bwd.py 22000 800 (each seed also: two launches bit-equal)"; timeout -k 10 400 python scratch/fuzz_bwd.py 22000 800 2>&1 | grep -v amdgpu.ids | tail -3
echo "## fuzz_bwd.py 23000 140 long"; timeout -k 10 300 python scratch/fuzz_bwd.py 23000 140 long 2>&1 | grep -v amdgpu.ids | tail -3
echo "## fuzz_fwd.py 90000 400 (product library)"; timeout -k 10 400 python scratch/fuzz_fwd.py 90000 400 2>&1 | grep -v amdgpu.ids | tail -3
} > gpurun_out/r05_fuzz_campaign.txt 2>&1
cat gpurun_out/r05_fuzz_campaign.txt
bash scratch/pmc_train.sh r05 > gpurun_out/r05_pmc_train.log 2>&1; echo "pmc_train rc=$?"
timeout -k 10 1000 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r05_gputests_final_tail.txt
echo "pytest rc=$?"; tail -4 gpurun_out/r05_gputests_final_tail.txt
