#!/bin/bash
# item 4 of the round-3 verdict: eight queries per wave (one wave per SIMD, query fragments in AGPRs) against the shipped
# four-per-wave instance, one process, same box; socket power sampled beside a sustained run of each
mkdir -p gpurun_out/r04
python scratch/variant_ab.py 1024 20000 6 0,60 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/q8_ab.txt
python scratch/variant_ab.py 256 20000 6 0,60 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/q8_ab.txt
python scratch/variant_ab.py 32 40000 6 0,60 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/q8_ab.txt
