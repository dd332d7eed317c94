// Corpus preparation kernels: HBM-bound byte work either side of the MaxSim loop.
//   pack_pmask : (np, lp) byte mask -> 32-patch tile words + per-page flags     (reads lp B / page)
//   split_f32  : fp32 rows -> bf16 hi/mid/lo planes, x == hi + mid + lo          (reads 512 B, writes 768 B / row)
// Mask semantics follow evaluator/retrieval.py:179-180,192,198 (mask.bool(), doc_has_token, -1e4 fill).
#include "evdr_common.h"

namespace {

__global__ void __launch_bounds__(64) pack_pmask_kernel(const uint8_t* __restrict__ pmask, int np, int lp, int ntiles,
                                                        uint32_t* __restrict__ tilemask,
                                                        uint32_t* __restrict__ pageflags) {
    const int page = blockIdx.x;
    const int lane = threadIdx.x;
    const uint8_t* row = pmask ? pmask + (int64_t)page * lp : nullptr;
    uint32_t any_valid = 0, first_masked = 0xFFFFu, nvalid = 0;
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
        any_valid |= w;
        nvalid += (uint32_t)__builtin_popcount(w);
        const uint32_t masked = inrange & ~w;
        if (masked) first_masked = min(first_masked, (uint32_t)(base + __builtin_ctz(masked)));
    }
    // wave reductions
    for (int off = 32; off > 0; off >>= 1) {
        any_valid |= __shfl_xor(any_valid, off);
        first_masked = min(first_masked, (uint32_t)__shfl_xor((int)first_masked, off));
        nvalid += (uint32_t)__shfl_xor((int)nvalid, off);
    }
    if (lane == 0) {
        uint32_t f = (any_valid ? 1u : 0u);
        if (first_masked != 0xFFFFu) f |= 2u | (first_masked << 16);
        // bit2: the valid patches are exactly the prefix [0, first_masked) (or the whole page): the tile masks are then a
        // function of that length alone and the forward kernel derives them without loading the mask words
        if (first_masked == 0xFFFFu || nvalid == first_masked) f |= 4u;
        pageflags[page] = f;
    }
}

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float x) {
    return __builtin_bit_cast(uint16_t, (__bf16)x);   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
}
__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }

// one thread = 8 consecutive floats (two 16-B loads, three 16-B stores)
__global__ void __launch_bounds__(256) split_f32_kernel(const float* __restrict__ x, int64_t n8,
                                                        uint16_t* __restrict__ hi, uint16_t* __restrict__ mid,
                                                        uint16_t* __restrict__ lo) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + i * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
        typedef __attribute__((ext_vector_type(8))) uint16_t u16x8;
        u16x8 a, b, c;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float f = (k < 4) ? v0[k & 3] : v1[k & 3];
            const uint16_t bh = f32_to_bf16_rne(f);
            const float r1 = f - bf16_to_f32(bh);          // exact in fp32
            const uint16_t bm = f32_to_bf16_rne(r1);
            const float r2 = r1 - bf16_to_f32(bm);         // exact in fp32
            a[k] = bh;
            b[k] = bm;
            c[k] = f32_to_bf16_rne(r2);
        }
        *reinterpret_cast<u16x8*>(hi + i * 8) = a;
        *reinterpret_cast<u16x8*>(mid + i * 8) = b;
        *reinterpret_cast<u16x8*>(lo + i * 8) = c;
    }
}

}  // namespace

hipError_t evdr_launch_pack_pmask(const uint8_t* pmask, int64_t np, int64_t lp, uint32_t* tilemask,
                                  uint32_t* pageflags, hipStream_t stream) {
    const int ntiles = (int)((lp + EVDR_TILE_PATCHES - 1) / EVDR_TILE_PATCHES);
    hipLaunchKernelGGL(pack_pmask_kernel, dim3((unsigned)np), dim3(64), 0, stream, pmask, (int)np, (int)lp, ntiles,
                       tilemask, pageflags);
    return hipGetLastError();
}

hipError_t evdr_launch_split_f32(const float* x, int64_t rows, uint16_t* planes, hipStream_t stream) {
    const int64_t n8 = rows * (EVDR_D / 8);
    if (n8 == 0) return hipSuccess;
    int64_t blocks = (n8 + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;     // grid-stride beyond 8 blocks per CU
    const int64_t plane = rows * EVDR_D;
    hipLaunchKernelGGL(split_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n8, planes, planes + plane,
                       planes + 2 * plane);
    return hipGetLastError();
}
