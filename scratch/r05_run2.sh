#!/bin/bash
# round 5, GPU call: the whole GPU suite (no -x: all failures at once), then the PMC passes over the training step
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r05_gputests_2.txt
echo "pytest rc=$?"; tail -4 gpurun_out/r05_gputests_2.txt
bash scratch/pmc_train.sh r05 > gpurun_out/r05_pmc_train.log 2>&1; echo "pmc_train rc=$?"
tail -30 gpurun_out/r05_pmc_train.log
