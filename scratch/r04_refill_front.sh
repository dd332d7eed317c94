#!/bin/bash
# A/B of the in-block refill placement: product (pieces of the next stage evenly over the block) vs -DEVDR_REFILL_FRONT=1 (over its first half)
{
echo "## headline shape, bf16, 1024 x 20000, interleaved"; python scratch/lib_ab.py 1024 20000 6 default scratch/ab/libevdr_front.so 2>&1 | grep -v amdgpu.ids
echo "## 256 x 20000 bf16"; python scratch/lib_ab.py 256 20000 8 default scratch/ab/libevdr_front.so 2>&1 | grep -v amdgpu.ids
echo "## 32 x 40000 bf16"; python scratch/lib_ab.py 32 40000 10 default scratch/ab/libevdr_front.so 2>&1 | grep -v amdgpu.ids
echo "## teacher shape, fp32 planes, 32 x 500, interleaved"; AB_DTYPE=f32 python scratch/lib_ab.py 32 500 60 default scratch/ab/libevdr_front.so 2>&1 | grep -v amdgpu.ids
echo "## fp32 planes, 500 x 6847"; AB_DTYPE=f32 python scratch/lib_ab.py 500 6847 6 default scratch/ab/libevdr_front.so 2>&1 | grep -v amdgpu.ids
echo "## student decomposition, product"; python scratch/student_decompose.py 2>&1 | grep -v amdgpu.ids
echo "## student decomposition, front"; EVDR_AB_LIB=scratch/ab/libevdr_front.so python scratch/student_decompose.py 2>&1 | grep -v amdgpu.ids
echo "## student decomposition, product again"; python scratch/student_decompose.py 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/r04_refill_front.txt
