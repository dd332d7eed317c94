#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
echo "# Round 5, after the deferred page-end stores of the fp16-plane forwards"
echo "## fuzz_fwd.py 91000 700 (product library)"; timeout -k 10 400 python scratch/fuzz_fwd.py 91000 700 2>&1 | grep -v amdgpu.ids | tail -2
echo "## fuzz_fwd.py 92000 300 through libevdr_sentinel.so"; EVDR_FUZZ_LIB=libevdr_sentinel.so timeout -k 10 400 python scratch/fuzz_fwd.py 92000 300 2>&1 | grep -v amdgpu.ids | tail -2
echo "## fuzz_fwd.py 93000 100 long"; timeout -k 10 400 python scratch/fuzz_fwd.py 93000 100 long 2>&1 | grep -v amdgpu.ids | tail -2
} > gpurun_out/r05_fuzz_campaign2.txt 2>&1
cat gpurun_out/r05_fuzz_campaign2.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r05_gputests_final_tail.txt
echo "pytest rc=$?"; tail -4 gpurun_out/r05_gputests_final_tail.txt
