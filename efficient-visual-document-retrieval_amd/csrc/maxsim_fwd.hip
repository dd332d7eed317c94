// MaxSim forward for gfx950 (MI355X): fused  Q·Pᵀ (bf16 MFMA, fp32 accumulate)  ->  masked max over a
// page's patches  ->  masked sum over query tokens.   Replaces evaluator/retrieval.py:187-211 of the
// reference (einsum -> masked_fill -> max -> *has -> *qmask -> sum over a materialised 4-D tensor).
//
// Mapping (MI355X-first, not a translation of the four ATen ops):
//   * one workgroup = 8 waves (2 per SIMD) = 8*QW queries  x  a chunk of consecutive pages;
//   * every wave keeps its QW queries' tokens RESIDENT in registers as the MFMA B operand
//     (token on the lane: C/D column = lane&31), so the max over patches is an in-register
//     v_max3 chain over the 16 accumulator registers + one cross-half exchange per page;
//   * pages stream HBM -> LDS with LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction)
//     through a 3-stage ring with counted vmcnt + raw s_barrier (one barrier per 4 tiles);
//     the LDS image is XOR-swizzled through the per-lane SOURCE address so that the
//     ds_read_b128 A-fragment reads are bank-conflict free;
//   * all 8 waves read the same patch tile from LDS: each page byte is fetched from HBM/L2 once per
//     8*QW queries, and the block->(query group, page chunk) map puts the query groups that share a
//     page chunk on one XCD so the chunk is served by that XCD's L2;
//   * NPL = 3 scores fp32 inputs to fp32 accuracy as six bf16 plane products (hi/mid/lo split).
#include <stdlib.h>

#include "maxsim_device.h"

namespace {

using namespace evdr;
constexpr int WAVES = kWaves;
constexpr int NSTAGE = 3;
constexpr int TILE_BYTES = kTileBytes;

// patch row inside a 32-patch tile held by accumulator register `reg` of lane half `h`
// (C/D map of mfma_f32_32x32x16: row = (reg&3) + 8*(reg>>2) + 4*h)
__host__ __device__ constexpr int acc_row(int reg) { return (reg & 3) + 8 * (reg >> 2); }

template <int QW, int NPL, bool ARGMAX>
__global__ void __launch_bounds__(WAVES * 64) maxsim_fwd_kernel(const EvdrFwdParams p) {
    constexpr int ST = (NPL == 1) ? 4 : 2;                 // tiles per ring stage
    constexpr int STAGE_BYTES = ST * NPL * TILE_BYTES;     // 32 KiB / 48 KiB
    constexpr int PIECES = ST * NPL * 8;                   // 1-KiB LDS-DMA pieces per stage
    constexpr int G = PIECES / WAVES;                      // pieces issued per wave per stage
    static_assert(PIECES % WAVES == 0, "stage must split evenly over the waves");
    // plane products (A = page plane, B = query plane), smallest magnitude first
    constexpr int NPROD = (NPL == 1) ? 1 : 6;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31;      // query token (MFMA column) / patch row for the A fragment
    const int h = lane >> 5;      // lane half

    const BlockWork bw = block_work(p);
    if (!bw.valid) return;
    const int qg = bw.qg, pg0 = bw.pg0, npages = bw.npages;
    const int total_tiles = npages * p.ntiles;
    const int nstages = (total_tiles + ST - 1) / ST;

    // ---- resident query fragments: B[k = 8h + j][col r] = Q[q][token r][16 ks + 8h + j]
    const int q0 = (qg * WAVES + wave) * QW;
    const bool active = q0 < p.nq;            // wave-uniform
    bf16x8 bq[QW][NPL][8];
    float qwt[QW];
#pragma unroll
    for (int j = 0; j < QW; ++j) {
        const int q = q0 + j;
        const bool ok = (q < p.nq) && (r < p.lq);
        const int64_t row = (int64_t)q * p.q_stride + (int64_t)(p.tok0 + r) * EVDR_D + h * 8;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) v = *reinterpret_cast<const bf16x8*>(p.Q + pl * p.q_plane_stride + row + ks * 16);
                bq[j][pl][ks] = v;
            }
        float w = 0.f;
        if (ok) w = (p.qmask == nullptr || p.qmask[(int64_t)q * p.lq_total + p.tok0 + r] != 0) ? 1.f : 0.f;
        qwt[j] = w;
    }
    // every ordinary load above is consumed before the first LDS-DMA is issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- LDS-DMA staging of ring stage s (tiles s*ST .. s*ST+ST-1 of this block's flat tile stream)
    const uint32_t smem_base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    auto issue_stage = [&](int s, int slot) {
        const uint32_t sbase = smem_base + slot * STAGE_BYTES;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int pc = wave * G + g;                    // wave-uniform piece id
            const int tis = pc / (8 * NPL);
            const int rem = pc - tis * (8 * NPL);
            const int pl = rem >> 3, piece = rem & 7;
            int t = s * ST + tis;
            t = min(t, total_tiles - 1);                    // tail: re-fetch the last tile (never read)
            const int pgi = t / p.ntiles;
            const int tip = t - pgi * p.ntiles;
            const int rit = piece * 4 + (lane >> 4);        // row inside the tile, 0..31
            const int row = min(tip * EVDR_TILE_PATCHES + rit, p.lp - 1);   // rows >= lp are masked: clamp
            const int csrc = (lane & 15) ^ (rit & 15);      // XOR swizzle lives on the SOURCE address
            const uint16_t* src = p.P + (int64_t)pl * p.p_plane_stride + (int64_t)(pg0 + pgi) * p.p_stride +
                                  (int64_t)row * EVDR_D + csrc * 8;
            const uint32_t dst = sbase + (tis * NPL + pl) * TILE_BYTES + piece * 1024;
            lds_dma_16B(src, __builtin_amdgcn_readfirstlane(dst));
        }
    };

    // ---- per-page running state
    float run[QW];
    int ridx[QW];
    int pgi = 0, tip = 0;
    // mask words are wave-uniform and written by an earlier launch: read them through the scalar cache
    // (constant address space) so they never enter the vmcnt queue that paces the LDS-DMA ring
    typedef const __attribute__((address_space(4))) uint32_t* cptr_t;
    cptr_t tilemask_c = (cptr_t)(uintptr_t)p.tilemask;
    cptr_t pageflags_c = (cptr_t)(uintptr_t)p.pageflags;
    uint32_t pflags = pageflags_c[pg0];
    auto reset_run = [&]() {
#pragma unroll
        for (int j = 0; j < QW; ++j) {
            // a masked patch inside [0, lp) puts -1e4 into the max (retrieval.py:185,198)
            run[j] = (pflags & 2u) ? -1e4f : neg_inf();
            ridx[j] = (int)(pflags >> 16);
        }
    };
    reset_run();

    const int hx = h ^ (r & 15);                          // swizzled 16-B chunk selector, see a-fragment read
    const int a_lane_off = r * (EVDR_D * 2);

    issue_stage(0, 0);
    if (nstages > 1) issue_stage(1, 1);
    int slot = 0;
    for (int s = 0; s < nstages; ++s) {
        // stage s has landed once all but the next stage's G pieces (and anything younger) are done
        if (s + 1 < nstages) wait_vmcnt<G>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (s + 2 < nstages) issue_stage(s + 2, slot == 0 ? 2 : slot - 1);
        const char* sbase = smem + slot * STAGE_BYTES;
        if (active) {
#pragma unroll
            for (int tis = 0; tis < ST; ++tis) {
                if (s * ST + tis < total_tiles) {
                    const int page = pg0 + pgi;
                    const uint32_t tm = tilemask_c[(int64_t)page * p.ntiles + tip];    // s_load_dword
                    if (tm != 0u) {
                        const char* tb = sbase + tis * NPL * TILE_BYTES + a_lane_off;
                        bf16x8 a[NPL][8];
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                            for (int ks = 0; ks < 8; ++ks)
                                a[pl][ks] = *reinterpret_cast<const bf16x8*>(tb + pl * TILE_BYTES + (((2 * ks) ^ hx) << 4));
                        const int pbase = tip * EVDR_TILE_PATCHES + 4 * h;
                        auto chain = [&](int j) {
                            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                            for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
                                for (int ks = 0; ks < 8; ++ks)
                                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                        a[NPL == 1 ? 0 : PA[pr]][ks], bq[j][NPL == 1 ? 0 : PB[pr]][ks], acc, 0, 0, 0);
                            return acc;
                        };
                        if (!ARGMAX && tm == 0xFFFFFFFFu) {
                            // every patch of the tile is valid (the common case): plain max tree
#pragma unroll
                            for (int j = 0; j < QW; ++j) {
                                const f32x16 acc = chain(j);
                                float m = acc[0];
#pragma unroll
                                for (int i = 1; i < 16; ++i) m = __builtin_fmaxf(m, acc[i]);
                                run[j] = __builtin_fmaxf(run[j], m);
                            }
                        } else {
                            uint32_t mybits = tm >> (4 * h);
                            asm volatile("" : "+v"(mybits));     // keep the bit tests inside this branch
#pragma unroll
                            for (int j = 0; j < QW; ++j) {
                                const f32x16 acc = chain(j);
                                if constexpr (!ARGMAX) {
                                    float m = neg_inf();
#pragma unroll
                                    for (int i = 0; i < 16; ++i)
                                        m = __builtin_fmaxf(m, ((mybits >> acc_row(i)) & 1u) ? acc[i] : neg_inf());
                                    run[j] = __builtin_fmaxf(run[j], m);
                                } else {
                                    float best = run[j];
                                    int bi = ridx[j];
#pragma unroll
                                    for (int i = 0; i < 16; ++i) {      // increasing patch order: first max wins
                                        const bool take = ((mybits >> acc_row(i)) & 1u) && (acc[i] > best);
                                        best = take ? acc[i] : best;
                                        bi = take ? (pbase + acc_row(i)) : bi;
                                    }
                                    run[j] = best;
                                    ridx[j] = bi;
                                }
                            }
                        }
                    }
                    // ---- page finished: fold halves, weight, reduce over tokens, store
                    if (++tip == p.ntiles) {
                        const float has = (pflags & 1u) ? 1.f : 0.f;       // doc_has_token (retrieval.py:192,204)
#pragma unroll
                        for (int j = 0; j < QW; ++j) {
                            float v = run[j];
                            const float o = __shfl_xor(v, 32);
                            if constexpr (ARGMAX) {
                                int bi = ridx[j];
                                const int oi = __shfl_xor(bi, 32);
                                const bool take = (o > v) || (o == v && oi < bi);
                                v = take ? o : v;
                                bi = take ? oi : bi;
                                if (h == 0 && r < p.lq && q0 + j < p.nq)
                                    p.argmax[((int64_t)(q0 + j) * p.np + page) * p.lq_total + p.tok0 + r] = (uint16_t)bi;
                            } else {
                                v = __builtin_fmaxf(v, o);
                            }
                            const float c = row32_sum(v * has * qwt[j]);
                            if (lane == 0 && q0 + j < p.nq) {
                                float* o = p.out + (int64_t)(q0 + j) * p.out_stride + page;
                                if (p.accumulate) atomicAdd(o, c);     // later 32-token slice: no-return atomic
                                else *o = c;
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

template <int QW, int NPL, bool ARGMAX>
hipError_t launch(const EvdrFwdParams& p, hipStream_t stream) {
    constexpr int ST = (NPL == 1) ? 4 : 2;
    constexpr int LDS = NSTAGE * ST * NPL * TILE_BYTES;
    auto kern = maxsim_fwd_kernel<QW, NPL, ARGMAX>;
    static uint64_t attr_devs = 0;
    if (hipError_t e = evdr_ensure_dyn_lds((const void*)kern, LDS, attr_devs); e != hipSuccess) return e;
    const int64_t blocks = (int64_t)((p.n_chunks + 7) / 8) * 8 * p.n_qgroups;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WAVES * 64), LDS, stream, p);
    return hipGetLastError();
}

}  // namespace

// Chooses queries-per-wave from the query count (more queries per wave = more reuse of each LDS read)
// and the pages-per-block chunk so that the grid is several times the CU count.
hipError_t evdr_launch_maxsim_fwd(const EvdrFwdParams& pin, int nplanes, bool want_argmax, hipStream_t stream) {
    EvdrFwdParams p = pin;
    p.ntiles = (p.lp + EVDR_TILE_PATCHES - 1) / EVDR_TILE_PATCHES;
    int qw = 1;
    if (nplanes == 1) qw = (p.nq > 16) ? 4 : (p.nq > 8 ? 2 : 1);
    if (nplanes == 1 && !want_argmax) {
        // bf16 scoring without argmax (retrieval / eval / teacher scores): the 16x16x32-shape kernel
        // (maxsim_fwd16.hip).  EVDR_FWD_VARIANT is an experiment switch read per launch:
        // 0/unset = default (staged kernel for long pages), 1 = flat per-tile kernel, 100 = this file's 32x32x16 kernel.
        const char* e = getenv("EVDR_FWD_VARIANT");
        const int variant = e ? atoi(e) : 0;
        if (variant != 100) return evdr_launch_maxsim_fwd16(p, qw, variant, stream);
    }
    evdr_set_geometry(p, WAVES * qw);
    if (nplanes == 3) {
        return want_argmax ? launch<1, 3, true>(p, stream) : launch<1, 3, false>(p, stream);
    }
    if (want_argmax) {
        if (qw == 4) return launch<4, 1, true>(p, stream);
        if (qw == 2) return launch<2, 1, true>(p, stream);
        return launch<1, 1, true>(p, stream);
    }
    if (qw == 4) return launch<4, 1, false>(p, stream);
    if (qw == 2) return launch<2, 1, false>(p, stream);
    return launch<1, 1, false>(p, stream);
}
