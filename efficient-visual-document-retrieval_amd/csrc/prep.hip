// Corpus preparation kernels: HBM-bound byte work either side of the MaxSim loop.
//   pack_pmask : (np, lp) byte mask -> 32-patch tile words + per-page flags     (reads lp B / page)
//   split_f32  : fp32 rows -> absmax, then fp16 hi/lo planes of x * 2^k          (reads 2 x 512 B, writes 512 B / row)
// Mask semantics follow evaluator/retrieval.py:179-180,192,198 (mask.bool(), doc_has_token, -1e4 fill).
#include "evdr_common.h"

namespace {

__global__ void __launch_bounds__(64) pack_pmask_kernel(const uint8_t* __restrict__ pmask, int np, int lp, int ntiles,
                                                        uint32_t* __restrict__ tilemask,
                                                        uint32_t* __restrict__ pageflags) {
    const int page = blockIdx.x;
    const int lane = threadIdx.x;
    const uint8_t* row = pmask ? pmask + (int64_t)page * lp : nullptr;
    uint32_t first_masked = 0xFFFFu, first_valid = 0xFFFFu, end_valid = 0u, nvalid = 0;
    for (int t = lane; t < ntiles; t += 64) {
        uint32_t w = 0, inrange = 0;
        const int base = t * 32;
#pragma unroll 8
        for (int m = 0; m < 32; ++m) {
            const int i = base + m;
            if (i < lp) {
                inrange |= 1u << m;
                if (row == nullptr || row[i] != 0) w |= 1u << m;
            }
        }
        tilemask[(int64_t)page * ntiles + t] = w;
        nvalid += (uint32_t)__builtin_popcount(w);
        if (w) {
            first_valid = min(first_valid, (uint32_t)(base + __builtin_ctz(w)));
            end_valid = max(end_valid, (uint32_t)(base + 32 - __builtin_clz(w)));
        }
        const uint32_t masked = inrange & ~w;
        if (masked) first_masked = min(first_masked, (uint32_t)(base + __builtin_ctz(masked)));
    }
    // wave reductions
    for (int off = 32; off > 0; off >>= 1) {
        first_masked = min(first_masked, (uint32_t)__shfl_xor((int)first_masked, off));
        first_valid = min(first_valid, (uint32_t)__shfl_xor((int)first_valid, off));
        end_valid = max(end_valid, (uint32_t)__shfl_xor((int)end_valid, off));
        nvalid += (uint32_t)__shfl_xor((int)nvalid, off);
    }
    if (lane == 0) {
        uint32_t f = (nvalid ? 1u : 0u);
        if (first_masked != 0xFFFFu) f |= 2u;
        // bit2: the valid patches are exactly ONE range [va, vb) (the whole page, a ragged prefix, an image between masked
        // text tokens, or nothing at all): va in bits 4..15, vb in bits 16..31.  The forward kernel then derives every tile
        // mask from the two numbers, walks only the stages inside the range and fetches only its tiles.  Otherwise (holes,
        // or va >= 4096) bits 16..31 hold the index of the first masked patch and the mask words are read.
        const uint32_t va = nvalid ? first_valid : 0u, vb = nvalid ? end_valid : 0u;
        if (nvalid == vb - va && va < 4096u) f |= 4u | (va << 4) | (vb << 16);
        else f |= (first_masked == 0xFFFFu ? 0u : first_masked) << 16;
        pageflags[page] = f;
    }
}

// absmax over the FINITE elements of the tensor, as raw bits (|x| as uint32 orders like the float).  NaN / Inf elements do
// not take part: one diverged row must not change the power-of-two scale of every other row.  They are reported instead:
// with `pageflags` given (pages: rows_per_page = lp, rowmask = pmask), a non-finite element in a VALID row sets bit 3 of
// its page's flag word, which makes the forward kernel return NaN for that page (include/evdr.h, "non-finite inputs").
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, int64_t n4, uint32_t* __restrict__ amax_bits,
                                                     const uint8_t* __restrict__ rowmask, int64_t rows_per_page,
                                                     uint32_t* __restrict__ pageflags) {
    uint32_t m = 0;
    auto fold = [&](const f32x4& v, int64_t i) {
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float f = v[k];      // by value: __builtin_bit_cast on the vector-element lvalue reads element 0 (hipcc 7.2)
            const uint32_t b = __builtin_bit_cast(uint32_t, f) & 0x7FFFFFFFu;
            if (b < 0x7F800000u) m = max(m, b); else bad = true;
        }
        if (bad && pageflags != nullptr) {                       // rare path: which row / page was that?
            const int64_t row = (i * 4) / EVDR_D;
            if (rowmask == nullptr || rowmask[row] != 0) atomicOr(&pageflags[row / rows_per_page], 8u);
        }
    };
    // four 16-B loads in flight per thread: with one, a 50-MB tensor took 21-23 us (2.3 TB/s, latency-bound: the grid
    // covers 4 MB per sweep); rocprofv3 of the reference's call pattern, profiles/r03_callpattern_kernel_stats.csv
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(x + (i + u * stride) * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) fold(v[u], i + u * stride);
    }
    for (; i < n4; i += stride) fold(*reinterpret_cast<const f32x4*>(x + i * 4), i);
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    __shared__ uint32_t wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    // one atomic per workgroup: thousands of atomics on one address serialise in L2 (40 us for a 50-MB tensor)
    if (threadIdx.x == 0) {
        m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (m != 0u) atomicMax(amax_bits, m);
    }
}

// Non-finite scan of 16-bit pages (a bf16 corpus, or the hi plane of fp16 hi/lo planes) and of fp32 pages that are not
// going through absmax_kernel: one 16-lane group per 128-wide row, 16 B (or 2 x 16 B) per lane; a NaN / Inf element in a
// valid row sets bit 3 of the page's flag word.  EXP16: exponent mask of the element type (bf16 0x7F80, fp16 0x7C00).
template <int KIND>    // 0 = fp32, 1 = bf16, 2 = fp16
__global__ void __launch_bounds__(256) nonfinite_scan_kernel(const void* __restrict__ x, const uint8_t* __restrict__ pmask,
                                                             int64_t np, int64_t lp, int64_t p_stride,
                                                             uint32_t* __restrict__ pageflags) {
    const int sub = threadIdx.x & 15;
    const int64_t rows = np * lp;
    for (int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); r < rows; r += (int64_t)gridDim.x * 16) {
        const int64_t page = r / lp, patch = r - page * lp;
        const int64_t off = page * p_stride + patch * EVDR_D + sub * 8;
        bool bad = false;
        if constexpr (KIND == 0) {
            const uint32_t* w = reinterpret_cast<const uint32_t*>(reinterpret_cast<const float*>(x) + off);
            const uint4 a = *reinterpret_cast<const uint4*>(w), b = *reinterpret_cast<const uint4*>(w + 4);
            const uint32_t e[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) bad |= (e[k] & 0x7F800000u) == 0x7F800000u;
        } else {
            constexpr uint32_t M = (KIND == 1) ? 0x7F80u : 0x7C00u;
            const uint4 a = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(x) + off);
            const uint32_t e[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) bad |= ((e[k] & M) == M) || ((e[k] & (M << 16)) == (M << 16));
        }
        if (bad && (pmask == nullptr || pmask[r] != 0)) atomicOr(&pageflags[page], 8u);
    }
}

// one thread = 8 consecutive floats (two 16-B loads, two 16-B stores): xs = x * 2^k, hi = fp16(xs), lo = fp16(xs - hi)
__global__ void __launch_bounds__(256) split_h2_kernel(const float* __restrict__ x, int64_t n8, const uint32_t* __restrict__ amax_bits,
                                                       _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
    const int k = evdr_h2_shift(*amax_bits);
    typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + i * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
        f16x8 a, b;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = __builtin_ldexpf((j < 4) ? v0[j & 3] : v1[j & 3], k);    // exact
            const _Float16 h = (_Float16)f;                                          // RNE
            a[j] = h;
            b[j] = (_Float16)(f - (float)h);                                         // the difference is exact in fp32
        }
        *reinterpret_cast<f16x8*>(hi + i * 8) = a;
        *reinterpret_cast<f16x8*>(lo + i * 8) = b;
    }
}

// Small tensors (a query batch: 32 x 32 x 128 fp32 = 512 KB): absmax and split in ONE launch.  Every workgroup reduces the
// WHOLE tensor for itself (it sits in L2 after the first reader; 16 workgroups x 0.5 MB is nothing) and then splits its own
// 8192 floats: no zeroing of the absmax word, no atomics, no second and third launch -- three launches in a row cost a
// training step 15 us of mostly launch latency right in front of its first big kernel.
__global__ void __launch_bounds__(1024) split_small_kernel(const float* __restrict__ x, int64_t n8, uint32_t* __restrict__ amax_bits,
                                                           _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
    uint32_t m = 0;
    const int64_t n4 = n8 * 2;
    auto fold = [&](const f32x4& v) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float f = v[k];
            const uint32_t b = __builtin_bit_cast(uint32_t, f) & 0x7FFFFFFFu;
            if (b < 0x7F800000u) m = max(m, b);                  // non-finite elements do not set the scale (absmax_kernel)
        }
    };
    int64_t i = threadIdx.x;
    for (; i + 7 * 1024 < n4; i += 8 * 1024) {                   // eight 16-B loads in flight per thread
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(x + (i + u * 1024) * 4);
#pragma unroll
        for (int u = 0; u < 8; ++u) fold(v[u]);
    }
    for (; i < n4; i += 1024) fold(*reinterpret_cast<const f32x4*>(x + i * 4));
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    __shared__ uint32_t wmax[16];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    m = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) m = max(m, wmax[w]);
    if (blockIdx.x == 0 && threadIdx.x == 0) *amax_bits = m;
    const int k = evdr_h2_shift(m);
    typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
    const int64_t e = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if (e < n8) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + e * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + e * 8 + 4);
        f16x8 a, b;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = __builtin_ldexpf((j < 4) ? v0[j & 3] : v1[j & 3], k);
            const _Float16 h = (_Float16)f;
            a[j] = h;
            b[j] = (_Float16)(f - (float)h);
        }
        *reinterpret_cast<f16x8*>(hi + e * 8) = a;
        *reinterpret_cast<f16x8*>(lo + e * 8) = b;
    }
}

// The same for MANY small tensors laid end to end (the batches of a training epoch: segment s = rows [s R, s R + R) of x,
// the last one possibly short): blockIdx.y = segment, each workgroup reduces ITS segment's absmax and splits its own 8192
// floats of it.  Every segment keeps its own absmax word and power of two, so its planes are bit for bit what
// split_small_kernel gives for that batch alone; segment s's planes are one contiguous (2, rows_s, 128) block at
// planes + s * 2 * R * 128 (the layout evdr_maxsim_fwd_prepared takes for the queries).
__global__ void __launch_bounds__(1024) split_segments_kernel(const float* __restrict__ x, int64_t rows, int64_t seg_rows,
                                                              uint32_t* __restrict__ amax_bits, _Float16* __restrict__ planes) {
    const int64_t seg = blockIdx.y;
    const int64_t r0 = seg * seg_rows;
    const int64_t nrows = min(seg_rows, rows - r0);
    const float* xs = x + r0 * EVDR_D;
    const int64_t n8 = nrows * (EVDR_D / 8), n4 = n8 * 2;
    uint32_t m = 0;
    auto fold = [&](const f32x4& v) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float f = v[k];
            const uint32_t b = __builtin_bit_cast(uint32_t, f) & 0x7FFFFFFFu;
            if (b < 0x7F800000u) m = max(m, b);                  // non-finite elements do not set the scale (absmax_kernel)
        }
    };
    int64_t i = threadIdx.x;
    for (; i + 7 * 1024 < n4; i += 8 * 1024) {                   // eight 16-B loads in flight per thread
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(xs + (i + u * 1024) * 4);
#pragma unroll
        for (int u = 0; u < 8; ++u) fold(v[u]);
    }
    for (; i < n4; i += 1024) fold(*reinterpret_cast<const f32x4*>(xs + i * 4));
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    __shared__ uint32_t wmax[16];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    m = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) m = max(m, wmax[w]);
    if (blockIdx.x == 0 && threadIdx.x == 0) amax_bits[seg] = m;
    const int k = evdr_h2_shift(m);
    typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
    _Float16* hi = planes + seg * 2 * seg_rows * EVDR_D;
    _Float16* lo = hi + nrows * EVDR_D;
    const int64_t e = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if (e < n8) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(xs + e * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(xs + e * 8 + 4);
        f16x8 a, b;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = __builtin_ldexpf((j < 4) ? v0[j & 3] : v1[j & 3], k);
            const _Float16 h = (_Float16)f;
            a[j] = h;
            b[j] = (_Float16)(f - (float)h);
        }
        *reinterpret_cast<f16x8*>(hi + e * 8) = a;
        *reinterpret_cast<f16x8*>(lo + e * 8) = b;
    }
}

// Stable compaction of the queries that have a valid token in [tok0, tok0 + 32): one workgroup, 256 queries per round.
__global__ void __launch_bounds__(256) build_qlist_kernel(const uint8_t* __restrict__ qmask, int nq, int lq, int tok0,
                                                         int32_t* __restrict__ qlist, int32_t* __restrict__ qcount) {
    __shared__ int wave_cnt[4];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    const int hi = min(lq, tok0 + 32);
    for (int q0 = 0; q0 < nq; q0 += 256) {
        const int q = q0 + tid;
        bool live = false;
        if (q < nq)
            for (int t = tok0; t < hi; ++t) live |= qmask[(int64_t)q * lq + t] != 0;
        const unsigned long long bal = __ballot(live);
        if (lane == 0) wave_cnt[wv] = __popcll(bal);
        __syncthreads();
        int before = base;
        for (int w = 0; w < wv; ++w) before += wave_cnt[w];
        if (live) qlist[before + __popcll(bal & ((1ull << lane) - 1ull))] = q;
        __syncthreads();
        if (tid == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (tid == 0) *qcount = base;
}

}  // namespace

hipError_t evdr_launch_pack_pmask(const uint8_t* pmask, int64_t np, int64_t lp, uint32_t* tilemask,
                                  uint32_t* pageflags, hipStream_t stream) {
    const int ntiles = (int)((lp + EVDR_TILE_PATCHES - 1) / EVDR_TILE_PATCHES);
    hipLaunchKernelGGL(pack_pmask_kernel, dim3((unsigned)np), dim3(64), 0, stream, pmask, (int)np, (int)lp, ntiles,
                       tilemask, pageflags);
    return hipGetLastError();
}

hipError_t evdr_launch_build_qlist(const uint8_t* qmask, int64_t nq, int64_t lq, int64_t tok0, int32_t* qlist, int32_t* qcount,
                                   hipStream_t stream) {
    hipLaunchKernelGGL(build_qlist_kernel, dim3(1), dim3(256), 0, stream, qmask, (int)nq, (int)lq, (int)tok0, qlist, qcount);
    return hipGetLastError();
}

hipError_t evdr_launch_flag_nonfinite(const void* P, int kind, const uint8_t* pmask, int64_t np, int64_t lp, int64_t p_stride,
                                      uint32_t* pageflags, hipStream_t stream) {
    const int64_t rows = np * lp;
    if (rows == 0) return hipSuccess;
    int64_t blocks = (rows + 15) / 16;
    if (blocks > 256 * 8) blocks = 256 * 8;
    auto kern = kind == 0 ? nonfinite_scan_kernel<0> : (kind == 1 ? nonfinite_scan_kernel<1> : nonfinite_scan_kernel<2>);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, stream, P, pmask, np, lp, p_stride, pageflags);
    return hipGetLastError();
}

// segments of `seg_rows` rows each (<= 2048: one workgroup reads a whole segment for its absmax), the last one short
hipError_t evdr_launch_split_f32_segments(const float* x, int64_t rows, int64_t seg_rows, uint16_t* planes, uint32_t* amax_bits,
                                          hipStream_t stream) {
    if (rows == 0) return hipSuccess;
    const int64_t nseg = (rows + seg_rows - 1) / seg_rows;
    const int64_t n8 = seg_rows * (EVDR_D / 8);
    hipLaunchKernelGGL(split_segments_kernel, dim3((unsigned)((n8 + 1023) / 1024), (unsigned)nseg), dim3(1024), 0, stream, x, rows, seg_rows,
                       amax_bits, (_Float16*)planes);
    return hipGetLastError();
}

hipError_t evdr_launch_split_f32(const float* x, int64_t rows, uint16_t* planes, uint32_t* amax_bits, hipStream_t stream) {
    return evdr_launch_split_f32_pages(x, rows, planes, amax_bits, nullptr, 1, nullptr, stream);
}

// the same for pages: non-finite elements of valid patches are reported in `pageflags` (bit 3) on the way
hipError_t evdr_launch_split_f32_pages(const float* x, int64_t rows, uint16_t* planes, uint32_t* amax_bits, const uint8_t* rowmask,
                                       int64_t rows_per_page, uint32_t* pageflags, hipStream_t stream) {
    const int64_t n8 = rows * (EVDR_D / 8);
    if (pageflags == nullptr && n8 > 0 && n8 <= 32768) {                 // <= 1 MB and nothing to report per page: one launch
        hipLaunchKernelGGL(split_small_kernel, dim3((unsigned)((n8 + 1023) / 1024)), dim3(1024), 0, stream, x, n8, amax_bits,
                           (_Float16*)planes, (_Float16*)planes + rows * EVDR_D);
        return hipGetLastError();
    }
    hipError_t e = hipMemsetAsync(amax_bits, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    if (n8 == 0) return hipSuccess;
    int64_t blocks = (n8 + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;     // grid-stride beyond 8 blocks per CU
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, stream, x, n8 * 2, amax_bits,
                       rowmask, rows_per_page, pageflags);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    const int64_t plane = rows * EVDR_D;
    hipLaunchKernelGGL(split_h2_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n8, amax_bits, (_Float16*)planes,
                       (_Float16*)planes + plane);
    return hipGetLastError();
}
