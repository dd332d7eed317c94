// InfoNCE-distillation pieces shared by the loss kernel (maxsim_bwd.hip) and the two kernels that can take its work over
// (the student forward's last workgroup, maxsim_fwd16.hip; the fused update's head, maxsim_bwd.hip):
// criterion.py:56-68 of the reference -- CE(score_s / temperature, argmax_p score_t), mean over the batch.
#pragma once
#include "evdr_common.h"

namespace evdr {

// order key of a float for arg-max purposes: larger key = larger value, -0.0 == +0.0, every NaN = the greatest key.  Key 0
// is below every real value (-inf maps to 0x007FFFFF + 1 ... > 0), so "no element seen" never wins.
__device__ __forceinline__ uint32_t nan_max_key(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0xFFFFFFFFu;      // NaN (integer test: the build assumes no NaNs in float compares)
    if (u == 0x80000000u) u = 0u;                                 // -0.0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// R rows of infonce_row_kernel (maxsim_bwd.hip) computed by ONE wave, with that kernel's bits: per row the wave plays all four waves of
// the kernel's 256-thread workgroup -- lane l holds the elements of the threads l, l + 64, l + 128, l + 192 --, every "thread" adds
// its elements in ascending order, every played wave is reduced by the same xor butterfly, and the four wave sums are added left to
// right: the additions that decide the bits of the sum happen in the kernel's own order (max and arg-max do not depend on the order).
// The R rows are independent and written phase by phase, so that their loads, divisions, exps and butterflies overlap (one wave per
// SIMD has nothing else to hide their latencies behind).  n <= 1024 (four elements per played thread, the kernel's in-register part).
// s is read past this CU's vector L1 (AGENT-scope loads): the caller may be the last workgroup of the very launch that wrote it.
// rows >= nrows are skipped (their outputs are undefined).  rl[r] = lse - s[tidx] / temp.
template <int R>
__device__ __forceinline__ void infonce_rows_by_wave(const float* const (&s)[R], const float* const (&t)[R], int nrows, int n, float temp,
                                                     float (&smax_out)[R], float (&sum_out)[R], int (&tidx_out)[R], float (&rl)[R]) {
    const int lane = threadIdx.x & 63;
    float z[R][4][4];
    float tv[R][4][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = lane + 64 * w + 256 * k;
                const bool ok = r < nrows && i < n;
                z[r][w][k] = ok ? __hip_atomic_load(s[r] + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
                tv[r][w][k] = ok ? t[r][i] : 0.f;
            }
    uint32_t tbest[R];
    int tidx[R];
    float smax[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        tbest[r] = 0u; tidx[r] = 0; smax[r] = -__builtin_inff();
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = lane + 64 * w + 256 * k;
                z[r][w][k] = (i < n) ? z[r][w][k] / temp : -__builtin_inff();
                if (i < n) {
                    const uint32_t tk = nan_max_key(tv[r][w][k]);
                    if (tk > tbest[r] || (tk == tbest[r] && i < tidx[r])) { tbest[r] = tk; tidx[r] = i; }
                    smax[r] = fmaxf(smax[r], z[r][w][k]);
                }
            }
    }
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t ov = (uint32_t)__shfl_xor((int)tbest[r], o);
            const int oi = __shfl_xor(tidx[r], o);
            if (ov > tbest[r] || (ov == tbest[r] && oi < tidx[r])) { tbest[r] = ov; tidx[r] = oi; }
            smax[r] = fmaxf(smax[r], __shfl_xor(smax[r], o));
        }
    }
    float part[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (tidx[r] < 0 || tidx[r] >= n) tidx[r] = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float e = expf(z[r][w][k] - smax[r]);
                if (lane + 64 * w + 256 * k < n) sum += e;
            }
            part[r][w] = sum;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {                   // wave_sum of every played wave of every row, level by level
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int w = 0; w < 4; ++w) part[r][w] += __shfl_xor(part[r][w], o);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float sum = part[r][0] + part[r][1] + part[r][2] + part[r][3];
        smax_out[r] = smax[r];
        sum_out[r] = sum;
        tidx_out[r] = tidx[r];
        const float lse = smax[r] + logf(sum);
        const float st = r < nrows ? __hip_atomic_load(s[r] + tidx[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        rl[r] = lse - st / temp;
    }
}

// mean of b row losses (b <= 1024, in LDS or global memory the caller may read plainly) in mean_kernel's / the ticket path's order,
// by ONE wave playing that kernel's 256 threads; the result is in every lane
__device__ __forceinline__ float infonce_mean_by_wave(const float* rl, int b) {
    const int lane = threadIdx.x & 63;
    float part[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        float v = 0.f;
        for (int i = lane + 64 * w; i < b; i += 256) v += rl[i];
        part[w] = wave_sum(v);
    }
    return (part[0] + part[1] + part[2] + part[3]) / (float)b;
}

// d loss / d score_s of one entry: (softmax - onehot) / (temp * B), from e = exp(s / temp - row max) and 1 / (row sum).  ONE spelling
// (an explicit fma) for the loss kernel and for the fused update's head, so that the two can never differ by a contraction choice.
__device__ __forceinline__ float infonce_grad_from_e(float e, float inv_sum, bool target, float inv_tb) {
    return __builtin_fmaf(e, inv_sum, target ? -1.f : -0.f) * inv_tb;
}
// ... from the row's statistics (the fused update's head)
__device__ __forceinline__ float infonce_grad_entry(float s, float temp, float smax, float sum, int tidx, int page, float inv_tb) {
    const float e = expf(s / temp - smax);
    const float inv_sum = 1.f / sum;
    return infonce_grad_from_e(e, inv_sum, page == tidx, inv_tb);
}

}  // namespace evdr
