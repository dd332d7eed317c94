// MaxSim forward on the 16x16x32 bf16 MFMA shape (bf16 retrieval hot path: NPL = 1, no argmax).
//
// Same structure as maxsim_fwd.hip (queries resident as the MFMA B operand, token on the lane, LDS-DMA ring,
// XOR-swizzled conflict-free A reads, XCD-aware block map, fast path for all-valid tiles).  Why a second
// shape: under MFMA-dense load on random data MI355X lowers its clock, and the clock it holds depends on the
// MFMA shape (MI355X_MICROARCH.md "DVFS give-back" item 7: the 16x16x32 loop delivers more FLOP/s than the
// 32x32x16 loop at equal cycles per FLOP).  The 16-patch granularity also trims the tail of a 1030-patch page
// (65 x 16 = 1040 rows instead of 33 x 32 = 1056).
//
// Fragment maps (cdna_hip_programming.md §3), lane l: c = l & 15, g = l >> 4
//   A[row c][k = 8g + j]   = patch (16u + c) of the tile, dims 32s + 8g + j      (u = 16-patch half, s = k-step)
//   B[k = 8g + j][col c]   = token (16t + c) of the query, dims 32s + 8g + j      (t = token half)
//   C/D[row 4g + reg][col c] -> lane holds, for token 16t + c, the patches 16u + 4g + reg, reg = 0..3
#include "evdr_common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4v;

namespace {

constexpr int TILE_BYTES = EVDR_TILE_PATCHES * EVDR_D * 2;      // 8 KiB

__device__ __forceinline__ float neg_inf() { return -__builtin_inff(); }

__device__ __forceinline__ void lds_dma_16B(const void* gsrc, uint32_t lds_base) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_base)
        : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// WAVES waves per workgroup, ST 32-patch tiles per ring stage, NSTAGE ring slots (NSTAGE-1 stages in flight
// beyond the one being computed).  Two geometries are used: 8 waves / 96 KiB (one workgroup per CU) and
// 4 waves / 64 KiB (two independent workgroups per CU, whose barriers do not couple the two waves of a SIMD).
template <int QW, int WAVES, int ST, int NSTAGE>
__global__ void __launch_bounds__(WAVES * 64, 2) maxsim_fwd16_kernel(const EvdrFwdParams p) {
    constexpr int STAGE_BYTES = ST * TILE_BYTES;
    constexpr int G = ST * 8 / WAVES;                           // LDS-DMA pieces per wave per stage
    static_assert((ST * 8) % WAVES == 0 && NSTAGE >= 2, "bad ring geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15;
    const int g = lane >> 4;

    const int b = blockIdx.x;
    const int xi = b >> 3;
    const int qg = xi % p.n_qgroups;
    const int chunk = (xi / p.n_qgroups) * 8 + (b & 7);
    if (chunk >= p.n_chunks) return;
    const int pg0 = chunk * p.pages_per_block;
    const int npages = min(p.pages_per_block, p.np - pg0);
    const int total_tiles = npages * p.ntiles;
    const int nstages = (total_tiles + ST - 1) / ST;

    // ---- resident query fragments: bq[j][t][s]
    const int q0 = (qg * WAVES + wave) * QW;
    const bool active = q0 < p.nq;
    bf16x8 bq[QW][2][4];
    float qwt[QW][2];
#pragma unroll
    for (int j = 0; j < QW; ++j) {
        const int q = q0 + j;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int tok = 16 * t + c;
            const bool ok = (q < p.nq) && (tok < p.lq);
            const int64_t row = (int64_t)q * p.q_stride + (int64_t)(p.tok0 + tok) * EVDR_D + g * 8;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) v = *reinterpret_cast<const bf16x8*>(p.Q + row + s * 32);
                bq[j][t][s] = v;
            }
            float w = 0.f;
            if (ok) w = (p.qmask == nullptr || p.qmask[(int64_t)q * p.lq_total + p.tok0 + tok] != 0) ? 1.f : 0.f;
            qwt[j][t] = w;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const uint32_t smem_base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    auto issue_stage = [&](int s, int slot) {
        const uint32_t sbase = smem_base + slot * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int pc = wave * G + i;
            const int tis = pc >> 3, piece = pc & 7;
            int t = s * ST + tis;
            t = min(t, total_tiles - 1);
            const int pgi = t / p.ntiles;
            const int tip = t - pgi * p.ntiles;
            const int rit = piece * 4 + (lane >> 4);
            const int row = min(tip * EVDR_TILE_PATCHES + rit, p.lp - 1);
            const int csrc = (lane & 15) ^ (rit & 15);
            const uint16_t* src = p.P + (int64_t)(pg0 + pgi) * p.p_stride + (int64_t)row * EVDR_D + csrc * 8;
            lds_dma_16B(src, __builtin_amdgcn_readfirstlane(sbase + tis * TILE_BYTES + piece * 1024));
        }
    };

    typedef const __attribute__((address_space(4))) uint32_t* cptr_t;
    cptr_t tilemask_c = (cptr_t)(uintptr_t)p.tilemask;
    cptr_t pageflags_c = (cptr_t)(uintptr_t)p.pageflags;

    float run[QW][2];
    int pgi = 0, tip = 0;
    uint32_t pflags = pageflags_c[pg0];
    auto reset_run = [&]() {
#pragma unroll
        for (int j = 0; j < QW; ++j) run[j][0] = run[j][1] = (pflags & 2u) ? -1e4f : neg_inf();
    };
    reset_run();

    // A fragment address: row (16u + c) * 256 + ((4s + g) ^ c) * 16   (row & 15 == c for both halves)
    const int a_lane_off = c * (EVDR_D * 2);
    const int gx = g ^ c;      // (4s + g) ^ c == (4s) ^ (g ^ c): 4s and g (< 4) share no bits, so 4s + g == 4s ^ g

#pragma unroll
    for (int i = 0; i < NSTAGE - 1; ++i)
        if (i < nstages) issue_stage(i, i);
    int slot = 0;
    for (int s = 0; s < nstages; ++s) {
        // stage s has landed once only the younger stages' pieces (NSTAGE-2 of them in steady state) remain
        if (NSTAGE >= 3 && s + 1 < nstages) {
            if (NSTAGE >= 4 && s + 2 < nstages) wait_vmcnt<2 * G>(); else wait_vmcnt<G>();
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        if (s + NSTAGE - 1 < nstages) issue_stage(s + NSTAGE - 1, slot == 0 ? NSTAGE - 1 : slot - 1);
        const char* sbase = smem + slot * STAGE_BYTES;
        if (active) {
#pragma unroll
            for (int tis = 0; tis < ST; ++tis) {
                if (s * ST + tis < total_tiles) {
                    const int page = pg0 + pgi;
                    const uint32_t tm = tilemask_c[(int64_t)page * p.ntiles + tip];
                    if (tm != 0u) {
                        const char* tb = sbase + tis * TILE_BYTES + a_lane_off;
                        const bool hi_live = (tm >> 16) != 0u;            // wave-uniform: skip an all-masked 16-patch half
                        bf16x8 a[2][4];
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) a[0][s4] = *reinterpret_cast<const bf16x8*>(tb + (((4 * s4) ^ gx) << 4));
                        if (hi_live) {
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4)
                                a[1][s4] = *reinterpret_cast<const bf16x8*>(tb + 16 * EVDR_D * 2 + (((4 * s4) ^ gx) << 4));
                        }
                        if (tm == 0xFFFFFFFFu) {
#pragma unroll
                            for (int j = 0; j < QW; ++j) {
#pragma unroll
                                for (int t = 0; t < 2; ++t) {
                                    f32x4v acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
                                    for (int s4 = 0; s4 < 4; ++s4) {
                                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][s4], bq[j][t][s4], acc0, 0, 0, 0);
                                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][s4], bq[j][t][s4], acc1, 0, 0, 0);
                                    }
                                    float m = run[j][t];
                                    m = __builtin_fmaxf(__builtin_fmaxf(m, acc0[0]), acc0[1]);
                                    m = __builtin_fmaxf(__builtin_fmaxf(m, acc0[2]), acc0[3]);
                                    m = __builtin_fmaxf(__builtin_fmaxf(m, acc1[0]), acc1[1]);
                                    m = __builtin_fmaxf(__builtin_fmaxf(m, acc1[2]), acc1[3]);
                                    run[j][t] = m;
                                }
                            }
                        } else {
                            uint32_t mybits = tm >> (4 * g);
                            asm volatile("" : "+v"(mybits));
#pragma unroll
                            for (int j = 0; j < QW; ++j) {
#pragma unroll
                                for (int t = 0; t < 2; ++t) {
                                    f32x4v acc0 = {0, 0, 0, 0};
#pragma unroll
                                    for (int s4 = 0; s4 < 4; ++s4)
                                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][s4], bq[j][t][s4], acc0, 0, 0, 0);
                                    float m = run[j][t];
#pragma unroll
                                    for (int i = 0; i < 4; ++i) m = __builtin_fmaxf(m, ((mybits >> i) & 1u) ? acc0[i] : neg_inf());
                                    if (hi_live) {
                                        f32x4v acc1 = {0, 0, 0, 0};
#pragma unroll
                                        for (int s4 = 0; s4 < 4; ++s4)
                                            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][s4], bq[j][t][s4], acc1, 0, 0, 0);
#pragma unroll
                                        for (int i = 0; i < 4; ++i)
                                            m = __builtin_fmaxf(m, ((mybits >> (16 + i)) & 1u) ? acc1[i] : neg_inf());
                                    }
                                    run[j][t] = m;
                                }
                            }
                        }
                    }
                    if (++tip == p.ntiles) {
                        const float has = (pflags & 1u) ? 1.f : 0.f;
#pragma unroll
                        for (int j = 0; j < QW; ++j) {
                            float cs = 0.f;
#pragma unroll
                            for (int t = 0; t < 2; ++t) {
                                float v = run[j][t];
                                v = __builtin_fmaxf(v, __shfl_xor(v, 16));
                                v = __builtin_fmaxf(v, __shfl_xor(v, 32));
                                cs += v * has * qwt[j][t];
                            }
                            cs += __shfl_xor(cs, 8);
                            cs += __shfl_xor(cs, 4);
                            cs += __shfl_xor(cs, 2);
                            cs += __shfl_xor(cs, 1);
                            if (lane == 0 && q0 + j < p.nq) {
                                float* o = p.out + (int64_t)(q0 + j) * p.out_stride + page;
                                if (p.accumulate) atomicAdd(o, cs);
                                else *o = cs;
                            }
                        }
                        tip = 0;
                        ++pgi;
                        if (pgi < npages) pflags = pageflags_c[pg0 + pgi];
                        reset_run();
                    }
                }
            }
        }
        slot = (slot == NSTAGE - 1) ? 0 : slot + 1;
    }
}


template <int QW, int WAVES, int ST, int NSTAGE>
hipError_t launch16(const EvdrFwdParams& pin, hipStream_t stream) {
    EvdrFwdParams p = pin;
    constexpr int LDS = NSTAGE * ST * TILE_BYTES;
    auto kern = maxsim_fwd16_kernel<QW, WAVES, ST, NSTAGE>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    p.ntiles = (p.lp + EVDR_TILE_PATCHES - 1) / EVDR_TILE_PATCHES;
    p.n_qgroups = (p.nq + WAVES * QW - 1) / (WAVES * QW);
    int64_t ppb = ((int64_t)p.np * p.n_qgroups) / 1536;
    if (ppb < 1) ppb = 1;
    if (ppb > 64) ppb = 64;
    p.pages_per_block = (int)ppb;
    p.n_chunks = (p.np + p.pages_per_block - 1) / p.pages_per_block;
    const int64_t blocks = (int64_t)((p.n_chunks + 7) / 8) * 8 * p.n_qgroups;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WAVES * 64), LDS, stream, p);
    return hipGetLastError();
}

}  // namespace

// geom: 0 = 8 waves, 4-tile stages, 3 slots (96 KiB, 1 workgroup/CU); 1 = 4 waves, 4-tile stages, 2 slots (64 KiB,
// 2 workgroups/CU); 2 = 4 waves, 2-tile stages, 4 slots (64 KiB, 2 workgroups/CU)
hipError_t evdr_launch_maxsim_fwd16(const EvdrFwdParams& p, int qw, int geom, hipStream_t stream) {
    if (geom == 1) {
        if (qw == 4) return launch16<4, 4, 4, 2>(p, stream);
        if (qw == 2) return launch16<2, 4, 4, 2>(p, stream);
        return launch16<1, 4, 4, 2>(p, stream);
    }
    if (geom == 2) {
        if (qw == 4) return launch16<4, 4, 2, 4>(p, stream);
        if (qw == 2) return launch16<2, 4, 2, 4>(p, stream);
        return launch16<1, 4, 2, 4>(p, stream);
    }
    if (qw == 4) return launch16<4, 8, 4, 3>(p, stream);
    if (qw == 2) return launch16<2, 8, 4, 3>(p, stream);
    return launch16<1, 8, 4, 3>(p, stream);
}
