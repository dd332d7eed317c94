#!/bin/bash
# cost of the construction-level ring hand-over (vmcnt + lgkmcnt(0) + s_barrier as one statement): round-3 library vs this one
# (and, when built, a variant without the lgkmcnt(0)), interleaved in one process on one box, at the HBM-bound, mixed and
# MFMA-bound ends
mkdir -p gpurun_out/r04
EXTRA=$(ls scratch/ab/libevdr_nolgkm.so 2>/dev/null)
for cfg in "1 40000 12" "8 40000 20" "6 40000 20" "12 40000 12" "32 40000 8" "1024 20000 5"; do
  set -- $cfg
  python scratch/lib_ab.py $1 $2 $3 scratch/ab/libevdr_r03.so default $EXTRA 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r04/ab_hardening.txt
