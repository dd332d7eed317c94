#!/bin/bash
# round-4 fuzz campaign: the rewritten (bit-reproducible) backward, and the forward through the product AND the sentinel build
mkdir -p gpurun_out/r04
{
echo "# Round 4 fuzz campaign, one gpurun call on 1 x MI355X"
echo "## fuzz_bwd.py 12000 800 (each seed also: two launches bit-equal)"; timeout -k 10 400 python scratch/fuzz_bwd.py 12000 800 2>&1 | grep -v amdgpu.ids | tail -3
echo "## fuzz_bwd.py 13000 140 long"; timeout -k 10 300 python scratch/fuzz_bwd.py 13000 140 long 2>&1 | grep -v amdgpu.ids | tail -3
echo "## fuzz_fwd.py 80000 900 (product library)"; timeout -k 10 400 python scratch/fuzz_fwd.py 80000 900 2>&1 | grep -v amdgpu.ids | tail -3
echo "## fuzz_fwd.py 81000 900 through libevdr_sentinel.so"; EVDR_FUZZ_LIB=libevdr_sentinel.so timeout -k 10 400 python scratch/fuzz_fwd.py 81000 900 2>&1 | grep -v amdgpu.ids | tail -3
echo "## fuzz_fwd.py 82000 200 long through libevdr_sentinel.so"; EVDR_FUZZ_LIB=libevdr_sentinel.so timeout -k 10 400 python scratch/fuzz_fwd.py 82000 200 long 2>&1 | grep -v amdgpu.ids | tail -3
} | tee gpurun_out/r04/fuzz_campaign.txt
