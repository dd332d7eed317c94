#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python scratch/bwd_lib_ab.py scratch/ab/libevdr_base.so scratch/ab/libevdr_bw_early1.so scratch/ab/libevdr_bw_early2.so > gpurun_out/r05_bwd_phaseA_ab2.txt 2>&1; echo "ab rc=$?"
cat gpurun_out/r05_bwd_phaseA_ab2.txt | cut -c1-330
