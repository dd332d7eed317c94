// Probe: sustained MFMA rate of gfx950 under its power limit, register operands only (no LDS, no HBM), for the two bf16
// shapes, with realistic operand data (unit-norm 128-dim Gaussian rows rounded to bf16) and with zeros.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, int NACC, int PAT = 0>
__global__ void __launch_bounds__(256) mfma_loop(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = src[(size_t)(tid & 4095) * 16 + j];
        b[j] = src[(size_t)(tid & 4095) * 16 + 8 + j];
    }
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[NACC];
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int q = 0; q < NACC; ++q)
                {
                    // PAT 0: A held for NACC MFMAs, B changes every MFMA (the production kernel's pattern);
                    // PAT 1: roles swapped; PAT 2: both change every MFMA; PAT 3: both held for NACC MFMAs
                    if constexpr (PAT == 0) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[(j + q) & 7], acc[q], 0, 0, 0);
                    if constexpr (PAT == 1) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[(j + q) & 7], a[j], acc[q], 0, 0, 0);
                    if constexpr (PAT == 2) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(j + q) & 7], b[(j + 2 * q + 1) & 7], acc[q], 0, 0, 0);
                    if constexpr (PAT == 3) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[j], acc[q], 0, 0, 0);
                }
        }
#pragma unroll
        for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    } else {
        f32x16 acc[NACC];
#pragma unroll
        for (int q = 0; q < NACC; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int q = 0; q < NACC; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[(j + q) & 7], acc[q], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < NACC; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[q][i];
    }
    out[tid] = s;
}

static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}

template <int SHAPE, int NACC, int PAT = 0>
static void run(const char* name, const bf16x8* src, float* out, int wgs, double secs_target) {
    const double flop_per_mfma = (SHAPE == 16) ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int iters = 2000;
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mfma_loop<SHAPE, NACC, PAT>), dim3(wgs), dim3(256), 0, 0, src, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)wgs * 4 * iters * 8 * NACC * flop_per_mfma;
        printf("%-28s wgs %4d iters %7d  %8.2f ms  %8.1f TFLOP/s\n", name, wgs, iters, ms, flops / (ms * 1e-3) / 1e12);
        fflush(stdout);
        if (rep == 0) iters = (int)(iters * (secs_target * 1e3 / ms));   // then ~secs_target per launch
    }
}

int main() {
    const size_t n = 4096 * 16;   // bf16x8 elements
    std::vector<uint16_t> h(n * 8);
    srand(1234);
    for (size_t r = 0; r < n * 8 / 128; ++r) {   // rows of 128: Gaussian, unit norm, bf16
        float v[128], nn = 0;
        for (int i = 0; i < 128; ++i) {
            float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = (rand() + 1.f) / (RAND_MAX + 2.f);
            v[i] = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
            nn += v[i] * v[i];
        }
        nn = 1.f / sqrtf(nn);
        for (int i = 0; i < 128; ++i) h[r * 128 + i] = f2bf(v[i] * nn);
    }
    bf16x8 *src, *zer;
    float* out;
    hipMalloc(&src, n * 16);
    hipMalloc(&zer, n * 16);
    hipMalloc(&out, 4096 * 256 * 4);
    hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    hipMemset(zer, 0, n * 16);
    const double T = 0.6;
    run<16, 8>("16x16x32 data  2w/SIMD", src, out, 512, T);
    run<32, 4>("32x32x16 data  2w/SIMD", src, out, 512, T);
    run<16, 8, 1>("16x16 data B-held A-rot", src, out, 512, T);
    run<16, 8, 2>("16x16 data both rotate", src, out, 512, T);
    run<16, 8, 3>("16x16 data both held", src, out, 512, T);
    run<16, 8, 0>("16x16x32 data again", src, out, 512, T);
    run<16, 8>("16x16x32 zeros 2w/SIMD", zer, out, 512, T);
    run<32, 4>("32x32x16 zeros 2w/SIMD", zer, out, 512, T);
    hipDeviceSynchronize();
    return 0;
}
