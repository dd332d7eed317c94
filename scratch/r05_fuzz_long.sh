#!/bin/bash
# round 5, final library: a long fuzz campaign on otherwise idle GPU minutes (progress lines keep the call alive)
mkdir -p gpurun_out
{
echo "# Round 5 long fuzz campaign on the final library, one gpurun call"
echo "## fuzz_fwd.py 100000 2500 (product library)"; timeout -k 10 500 python scratch/fuzz_fwd.py 100000 2500 2>&1 | grep -v amdgpu.ids | tail -2
echo "## fuzz_fwd.py 110000 1500 through libevdr_sentinel.so"; EVDR_FUZZ_LIB=libevdr_sentinel.so timeout -k 10 500 python scratch/fuzz_fwd.py 110000 1500 2>&1 | grep -v amdgpu.ids | tail -2
echo "## fuzz_bwd.py 120000 2500"; timeout -k 10 500 python scratch/fuzz_bwd.py 120000 2500 2>&1 | grep -v amdgpu.ids | tail -2
echo "## fuzz_topk.py"; timeout -k 10 200 python scratch/fuzz_topk.py 130000 400 2>&1 | grep -v amdgpu.ids | tail -2
} 2>&1 | tee gpurun_out/r05_fuzz_long.txt
