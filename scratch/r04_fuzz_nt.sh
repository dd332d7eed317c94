#!/bin/bash
# round-4 (second session): the forward fuzz with the non-temporal corpus stream of the fp16-plane forward FORCED ON (variant 33; the
# dispatch rule picks it only for corpora >= 128 MiB), through the product and the sentinel build
{
echo "# fuzz of the nt instances (evdr_debug_set_fwd_variant(33)), one gpurun call on 1 x MI355X"
echo "## fuzz_fwd.py 90000 600, variant 33 (product library)"; EVDR_FUZZ_VARIANT=33 timeout -k 10 400 python scratch/fuzz_fwd.py 90000 600 2>&1 | grep -v amdgpu.ids | tail -3
echo "## fuzz_fwd.py 90600 300, variant 33 through libevdr_sentinel.so"; EVDR_FUZZ_VARIANT=33 EVDR_FUZZ_LIB=libevdr_sentinel.so timeout -k 10 400 python scratch/fuzz_fwd.py 90600 300 2>&1 | grep -v amdgpu.ids | tail -3
echo "## fuzz_fwd.py 91000 60 long, variant 33 through libevdr_sentinel.so"; EVDR_FUZZ_VARIANT=33 EVDR_FUZZ_LIB=libevdr_sentinel.so timeout -k 10 400 python scratch/fuzz_fwd.py 91000 60 long 2>&1 | grep -v amdgpu.ids | tail -3
echo "## fuzz_bwd.py 14000 300 (final library; each seed also: two launches bit-equal)"; timeout -k 10 400 python scratch/fuzz_bwd.py 14000 300 2>&1 | grep -v amdgpu.ids | tail -3
} | tee gpurun_out/r04_fuzz_nt.txt
