#!/bin/bash
# round-4 regression artefacts on the final build: the driver end to end (600 steps, three modes, loss trajectories compared) and the
# headline shape under the page-mask layouts
mkdir -p gpurun_out/r04
timeout -k 10 500 python scratch/driver_soak.py 600 2>&1 | grep -v "amdgpu.ids\|\[save\]\|^\[20" | tail -6 | tee gpurun_out/r04/driver_soak.txt
timeout -k 10 300 python scratch/masked_layouts.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/masked_layouts.txt
