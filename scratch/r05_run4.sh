#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_sentinel.py tests/test_gpu_edge_semantics.py -q 2>&1 | tail -30 > gpurun_out/r05_gputests_3.txt
echo "pytest rc=$?"; tail -3 gpurun_out/r05_gputests_3.txt
timeout -k 10 500 python scratch/driver_soak_default.py 1200 > gpurun_out/r05_driver_soak.txt 2>&1; echo "soak rc=$?"; tail -45 gpurun_out/r05_driver_soak.txt | cut -c1-220
