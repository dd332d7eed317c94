// Probe (round 3): would a 32x32x16 MFMA instance lower the energy per FLOP of the 5-16-query regime?
// The regime is power-bound (profiles/r02_power_probe.txt), so under the 1400 W cap the shape that sustains MORE TFLOP/s
// at the SAME LDS->register traffic per FLOP is the one that costs fewer joules per FLOP.
//
// Both shapes are fed the way the scorer feeds them: the query fragments (B operand) stay in registers, the page
// fragments (A operand) come from LDS with one ds_read_b128 per k-step, accumulators start from zero for every tile and
// are folded into a running maximum afterwards (v_max3 tree), QW queries per wave.
//   16x16x32: a fragment = 16 patches x 32 k, feeds 2 * QW MFMAs of 16 384 FLOP (both token halves of QW queries)
//   32x32x16: a fragment = 32 patches x 16 k, feeds     QW MFMAs of 32 768 FLOP (all 32 tokens of QW queries)
// i.e. the SAME FLOP per LDS byte for a given QW -- the shapes differ in the matrix pipe and the accumulator file only.
// Data: unit-norm 128-dim Gaussian rows in bf16 (the bench's statistics).  No HBM traffic inside the timed loop.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_lds_shapes mfma_lds_shapes.hip ; run: ./mfma_lds_shapes
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <sys/time.h>

static double now() { timeval tv; gettimeofday(&tv, nullptr); return tv.tv_sec + tv.tv_usec * 1e-6; }

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int LDS_BYTES = 64 * 1024;          // per workgroup: 2 workgroups of 4 waves per CU, like the 5-12-query instances
constexpr int FRAGS = LDS_BYTES / 1024;       // 1-KiB fragments (one ds_read_b128 per wave)

template <int SHAPE, int QW>
__global__ void __launch_bounds__(256, 2) lds_fed(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
    __shared__ bf16x8 lds[LDS_BYTES / 16];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < LDS_BYTES / 16; i += 256) lds[i] = src[((size_t)blockIdx.x * 97 + i) & 65535];
    bf16x8 b[QW][8];
#pragma unroll
    for (int q = 0; q < QW; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) b[q][j] = src[((size_t)(blockIdx.x * 256 + tid) * 8 + q * 8 + j + 31) & 65535];
    __syncthreads();
    float best[QW];
#pragma unroll
    for (int q = 0; q < QW; ++q) best[q] = -1e30f;
    int frag = (tid >> 6) * 3;
    for (int it = 0; it < iters; ++it) {
        if constexpr (SHAPE == 16) {
            // one 32-patch tile = 2 half-tiles x 4 k-steps; per half-tile 2*QW chains of 4 dependent MFMAs
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x4 acc[QW][2];
                bf16x8 a[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] = lds[((frag + half * 4 + k) & (FRAGS - 1)) * 64 + lane];
#pragma unroll
                for (int q = 0; q < QW; ++q)
#pragma unroll
                    for (int th = 0; th < 2; ++th) {
                        acc[q][th] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[q][th * 4 + 0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                        for (int k = 1; k < 4; ++k)
                            acc[q][th] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k], b[q][th * 4 + k], acc[q][th], 0, 0, 0);
                    }
#pragma unroll
                for (int q = 0; q < QW; ++q)
#pragma unroll
                    for (int th = 0; th < 2; ++th)
                        best[q] = fmaxf(fmaxf(best[q], fmaxf(acc[q][th][0], acc[q][th][1])), fmaxf(acc[q][th][2], acc[q][th][3]));
            }
        } else {
            // one 32-patch tile = 8 k-steps of 16; per tile QW chains of 8 dependent MFMAs
            f32x16 acc[QW];
            bf16x8 a[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = lds[((frag + k) & (FRAGS - 1)) * 64 + lane];
#pragma unroll
            for (int q = 0; q < QW; ++q) {
                f32x16 z;
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = 0.f;
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[q][0], z, 0, 0, 0);
#pragma unroll
                for (int k = 1; k < 8; ++k) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[q][k], acc[q], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < QW; ++q) {
                float m = best[q];
#pragma unroll
                for (int i = 0; i < 16; i += 2) m = fmaxf(m, fmaxf(acc[q][i], acc[q][i + 1]));
                best[q] = m;
            }
        }
        frag = (frag + 8) & (FRAGS - 1);
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < QW; ++q) s += best[q];
    out[blockIdx.x * 256 + tid] = s;
}

static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}

template <int SHAPE, int QW>
static void run(const char* name, const bf16x8* src, float* out, double secs_target) {
    const int wgs = 512;
    const double flop_per_iter = 2.0 * 32 * 32 * 128 * QW;       // one 32-patch tile against QW queries of 32 tokens, per wave
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int iters = 2000;
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((lds_fed<SHAPE, QW>), dim3(wgs), dim3(256), 0, 0, src, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)wgs * 4 * iters * flop_per_iter;
        const double lds_bytes = (double)wgs * 4 * iters * 8 * 1024;
        printf("t=%.2f %-30s iters %8d  %8.2f ms  %8.1f TFLOP/s  LDS->reg %6.2f TB/s  (%.4f B/FLOP)\n", now(), name, iters, ms,
               flops / (ms * 1e-3) / 1e12, lds_bytes / (ms * 1e-3) / 1e12, lds_bytes / flops);
        fflush(stdout);
        if (rep == 0) iters = (int)(iters * (secs_target * 1e3 / ms));
    }
}

int main() {
    const size_t n = 65536;   // bf16x8 elements = 1 MiB
    std::vector<uint16_t> h(n * 8);
    srand(1234);
    for (size_t r = 0; r < n * 8 / 128; ++r) {
        float v[128], nn = 0;
        for (int i = 0; i < 128; ++i) {
            float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = (rand() + 1.f) / (RAND_MAX + 2.f);
            v[i] = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
            nn += v[i] * v[i];
        }
        nn = 1.f / sqrtf(nn);
        for (int i = 0; i < 128; ++i) h[r * 128 + i] = f2bf(v[i] * nn);
    }
    bf16x8* src;
    float* out;
    hipMalloc(&src, n * 16);
    hipMalloc(&out, 512 * 256 * 4);
    hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    const double T = 0.6;
    run<16, 1>("16x16x32  1 query/wave", src, out, T);
    run<32, 1>("32x32x16  1 query/wave", src, out, T);
    run<16, 2>("16x16x32  2 queries/wave", src, out, T);
    run<32, 2>("32x32x16  2 queries/wave", src, out, T);
    run<16, 4>("16x16x32  4 queries/wave", src, out, T);
    run<32, 4>("32x32x16  4 queries/wave", src, out, T);
    run<16, 1>("16x16x32  1 query/wave again", src, out, T);
    hipDeviceSynchronize();
    return 0;
}
