#!/bin/bash
# round 5, final library: stage-size A/B of the fp16-plane forwards after the deferred stores; bit-stability stress (every launch of
# 8 shapes compared bit for bit with the first of its shape, plain and with alternating query masks); the headline over 60 steps
mkdir -p gpurun_out
timeout -k 10 200 python scratch/student_stage_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_student_stage_ab.txt
{ timeout -k 10 400 python scratch/determinism_stress.py 25 2>&1 | grep -v amdgpu.ids | tail -12; timeout -k 10 300 python scratch/determinism_stress.py 15 0 masked 2>&1 | grep -v amdgpu.ids | tail -12; } | tee gpurun_out/r05_determinism_stress.txt
timeout -k 10 300 python bench.py --steps 60 --warmup 2 --no-cpu-baseline --no-extras --no-other-regimes 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('bench 60 steps:', round(d['value'] / 1e6, 2), 'M pairs/s', round(d['ms_per_step'], 2), 'ms/step  kernel', round(d['roofline']['kernel_ms'], 2), 'ms  frac', round(d['roofline']['frac'], 4), 'ndcg@5', d['ndcg_at_5'])" | tee gpurun_out/r05_soak.txt
