// Backward of MaxSim w.r.t. the page embeddings, and the InfoNCE-distillation loss with its gradient.
//
// Reference behaviour being replaced: autograd of evaluator/retrieval.py:195-210 materialises a dense
// one-hot (Q, C, Lq, Lp) gradient and runs a dense einsum over it (mainv2_iter_distill_infonce.py:290).
// Here the forward kernel recorded argmax[q,p,n]; the backward is a gather of <= B*Lq query-token rows
// per page, summed in LDS (ds_add_f32) and written once:  HBM/LDS-bound, no MFMA, no global atomics.
//   bytes per page: read 2*B*Lq (argmax) + <= B*Lq*512 (Q rows, L2-resident) ; write Lp*512 (dP).
#include "evdr_common.h"

namespace {

constexpr int BW_THREADS = 512;
#ifndef EVDR_BW_EARLY_ROW
#define EVDR_BW_EARLY_ROW 1       /* A/B: 0 = the epilogue's first row is loaded when the epilogue starts (the form until round 5); 1 = its parameter part in front of the gather; 2 = at the kernel's head */
#endif
#ifndef EVDR_BW_SMALL_WGS
#define EVDR_BW_SMALL_WGS 384     /* launches of at most this many 128-row workgroups take 64-row slabs (A/B builds: -DEVDR_BW_FORCE_CAP=64|128) */
#endif

// One workgroup owns up to BW_ROWS consecutive patch rows of one page (slab_rows of them: equally tall slabs per page).  fp32 LDS atomics are NOT the accumulation path:
// ds_add_f32 costs ~175 cycles per wave-instruction on gfx950 (measured: 346 us with one atomic per pair, 28 us with plain
// read-modify-write on this very kernel).  Instead the (query, token) pairs are bucketed by target row -- a STABLE counting
// sort: per (row, wave-instruction) counts, exclusive scan, scatter by rank: three small LDS passes per chunk of pairs -- and
// the sorted list is cut into equal slices (snapped to bucket starts), one per 16-lane group: a group walks its slice with
// four 512-B query-row loads in flight, accumulates a row in registers and adds it into the slab with plain LDS accesses;
// only a row heavier than a whole slice is shared between groups.  Every dP element is written to HBM exactly once; masked
// rows come out as exact zeros.
// BIT-REPRODUCIBLE (the reference's CPU backward is): the order of the additions into one dP row is a function of the
// inputs alone, never of timing --
//   * inside a bucket the pairs stand in ascending (query, token) order: a pair's slot is bucket start + pairs of the same
//     row in earlier wave-instructions (byte counters per (row, instruction): integer adds, order-free) + its rank among the
//     lanes of its own instruction with the same row (a lane match in registers: 7 ballots, v_mbcnt) -- no cursor atomics whose
//     return value depends on who came first;
//   * a row shared by several groups (heavier than a slice) is summed in GROUP order: each group keeps the partial sum of a
//     row it shares in registers and the groups add them to the slab in ordered rounds (round t = the t-th group of the
//     row, plain read-modify-write, one barrier per round; no round at all when no row is shared) -- no float atomics.
// FUSED: instead of writing dP (the gradient w.r.t. the NORMALISED pages), the epilogue continues on the slab in LDS:
// l2-normalise backward through y = m x / (||m x|| + eps_n)  ->  AdamW on the raw parameter x (torch semantics: decoupled
// weight decay, bias-corrected moments), in place on x / exp_avg / exp_avg_sq.  One kernel replaces maxsim_bwd + the
// normalise/mask backward + torch's five multi-tensor AdamW kernels, and dP never goes to HBM.
struct AdamArgs {
    float* x;            // (np, lp, 128) raw parameter, updated in place
    float* exp_avg;      // first moment
    float* exp_avg_sq;   // second moment
    // hyper-parameters arrive as doubles (Python floats) and are combined in double on the host like torch does:
    // decay = 1 - lr * weight_decay, w1 = 1 - beta1, w2 = 1 - beta2 (1 - float(0.999) would be off by 1.3e-5 relative)
    float lr, decay, w1, beta2, w2, eps, bc1, bc2_sqrt, eps_norm;
    const float* bc_dev; // null, or {bc1, bc2_sqrt} in device memory (step counted on the device: HIP-graph replays)
    // null, or where the NEXT forward's operands go: l2_normalize(m * x_new) as the scorer's fp16 hi/lo planes (exactly what
    // l2norm_fwd_kernel<true> would produce from the updated x), the absmax word of the planes and the page flag words that
    // collect non-finite parameters -- the step's separate normalise pass (a read of x and a 20-us launch) disappears
    _Float16* next_hi;
    _Float16* next_lo;
    uint32_t* next_amax;
    uint32_t* pageflags;
};

// Device-resident step counter of an AdamW state: {int64 step; float bc1 = 1 - beta1^step; float bc2_sqrt}.  One thread; part
// of the captured graph of a training step, so a replay needs no scalar from the host.
__global__ void adamw_advance_kernel(void* state, double beta1, double beta2) {
    long long* step = reinterpret_cast<long long*>(state);
    float* bc = reinterpret_cast<float*>(step + 1);
    const long long t = *step + 1;
    *step = t;
    bc[0] = (float)(1.0 - pow(beta1, (double)t));
    bc[1] = (float)sqrt(1.0 - pow(beta2, (double)t));
}

// torch.optim.AdamW's update (amsgrad = False, maximize = False) of one fp32 tensor as ONE pure stream: g, x, exp_avg, exp_avg_sq
// read once, x / exp_avg / exp_avg_sq written once (28 B per element; torch's default foreach form makes eight passes over the
// tensor: 207 us against 60 for the 13.5 M student parameters of the training step).  Same expressions in the same order as
// torch's single-tensor form: decay, lerp of the first moment, second moment, addcdiv with step_size = lr / bc1.
// n4 = elements / 4; tail elements (n % 4) are handled by the first threads.
__global__ void __launch_bounds__(256) adamw_kernel(const float* __restrict__ g, float* __restrict__ x, float* __restrict__ ea,
                                                    float* __restrict__ es, int64_t n, float decay, float w1, float beta2,
                                                    float w2, float step_size, float bc2_sqrt, float eps) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
        f32x4 av = reinterpret_cast<const f32x4*>(ea)[i];
        f32x4 sv = reinterpret_cast<const f32x4*>(es)[i];
        xv *= decay;
        av = av + (gv - av) * w1;
        sv = sv * beta2 + gv * gv * w2;
#pragma unroll
        for (int k = 0; k < 4; ++k) xv[k] -= step_size * (av[k] / (sqrtf(sv[k]) / bc2_sqrt + eps));
        reinterpret_cast<f32x4*>(x)[i] = xv;
        reinterpret_cast<f32x4*>(ea)[i] = av;
        reinterpret_cast<f32x4*>(es)[i] = sv;
    }
    const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        const float gv = g[t];
        float xv = x[t] * decay;
        const float av = ea[t] + (gv - ea[t]) * w1;
        const float sv = es[t] * beta2 + gv * gv * w2;
        xv -= step_size * (av / (sqrtf(sv) / bc2_sqrt + eps));
        x[t] = xv;
        ea[t] = av;
        es[t] = sv;
    }
}

template <int BW_ROWS, int CHUNK, bool FUSED>
__global__ void __launch_bounds__(BW_THREADS) maxsim_bwd_kernel(const float* __restrict__ g,
                                                               const float* __restrict__ Q,
                                                               const uint8_t* __restrict__ qmask,
                                                               const uint8_t* __restrict__ pmask,
                                                               const uint16_t* __restrict__ argmax,
                                                               float* __restrict__ dP, int nq, int lq, int np, int lp,
                                                               int slab_rows, AdamArgs ad) {
    constexpr int PER_THREAD = CHUNK / BW_THREADS;
    constexpr int NGROUPS = BW_THREADS / 16;
    constexpr int NWAVES = BW_THREADS / 64;
    constexpr int NSLOTS = PER_THREAD * NWAVES;                          // wave-instructions that bucket pairs, per chunk
    constexpr int CW = (NSLOTS + 3) / 4;                                 // counter words per row (one byte per slot: <= 64 lanes)
    static_assert(CHUNK % BW_THREADS == 0 && (BW_ROWS == 64 || BW_ROWS == 128), "bucket geometry: one or two whole waves of buckets");
    extern __shared__ __attribute__((aligned(16))) float acc[];          // [BW_ROWS][128]
    int* list = reinterpret_cast<int*>(acc + BW_ROWS * EVDR_D);          // [CHUNK] pair index inside the chunk, bucketed
    float* wl = reinterpret_cast<float*>(list + CHUNK);                  // [CHUNK] weight of list[k]
    int* hist = reinterpret_cast<int*>(wl + CHUNK);                      // [BW_ROWS] bucket sizes
    int* offs = hist + BW_ROWS;                                          // [BW_ROWS] bucket starts
    uint32_t* cnt = reinterpret_cast<uint32_t*>(offs + BW_ROWS);         // [BW_ROWS][CW] pairs per (row, slot), one byte each
    __shared__ int sh_has, sh_total, sh_rounds, sh_wsum[2];
    const int page = blockIdx.x;
    // BW_ROWS is the slab's CAPACITY (LDS); a workgroup owns slab_rows <= BW_ROWS rows, chosen per launch so that a page's slabs
    // are equally tall (206 patches: 103 + 103, not 128 + 78)
    const int r0 = blockIdx.y * slab_rows;
    const int rows = min(slab_rows, lp - r0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = tid >> 6;
    const int gid = tid >> 4, sub = tid & 15;

    // FUSED epilogue operands of one parameter row (one 16-lane group per row, 8 floats per lane)
    struct RowIn { f32x4 x0, x1, ea0, ea1, es0, es1; float m; };
    auto load_row = [&](int r) {
        RowIn in;
        const int64_t off = ((int64_t)page * lp + r0 + r) * EVDR_D + sub * 8;
        in.m = (pmask == nullptr || pmask[(int64_t)page * lp + r0 + r] != 0) ? 1.f : 0.f;
        in.x0 = *reinterpret_cast<const f32x4*>(ad.x + off);
        in.x1 = *reinterpret_cast<const f32x4*>(ad.x + off + 4);
        in.ea0 = *reinterpret_cast<const f32x4*>(ad.exp_avg + off);
        in.ea1 = *reinterpret_cast<const f32x4*>(ad.exp_avg + off + 4);
        in.es0 = *reinterpret_cast<const f32x4*>(ad.exp_avg_sq + off);
        in.es1 = *reinterpret_cast<const f32x4*>(ad.exp_avg_sq + off + 4);
        return in;
    };
    // the same in two halves: the parameter row (9 registers) can be requested while the gather still runs, the moments
    // (16 more) only when the gather's registers are free
    auto load_row_x = [&](RowIn& in, int r) {
        const int64_t off = ((int64_t)page * lp + r0 + r) * EVDR_D + sub * 8;
        in.m = (pmask == nullptr || pmask[(int64_t)page * lp + r0 + r] != 0) ? 1.f : 0.f;
        in.x0 = *reinterpret_cast<const f32x4*>(ad.x + off);
        in.x1 = *reinterpret_cast<const f32x4*>(ad.x + off + 4);
    };
    auto load_row_moments = [&](RowIn& in, int r) {
        const int64_t off = ((int64_t)page * lp + r0 + r) * EVDR_D + sub * 8;
        in.ea0 = *reinterpret_cast<const f32x4*>(ad.exp_avg + off);
        in.ea1 = *reinterpret_cast<const f32x4*>(ad.exp_avg + off + 4);
        in.es0 = *reinterpret_cast<const f32x4*>(ad.exp_avg_sq + off);
        in.es1 = *reinterpret_cast<const f32x4*>(ad.exp_avg_sq + off + 4);
    };
    RowIn nxt{};
    // Every global load of the bucketing phase is issued up front, in ONE latency period: the page's mask bytes (has(p) =
    // any(pmask[p]), retrieval.py:192) together with the first chunk's arg-max / upstream-gradient / query-mask words; the slab and the
    // counters are cleared while they fly.  (Until round 5 the mask scan, a barrier and only then the pair loads: two dependent round
    // trips to L2 at the head of every workgroup, with the HBM stream of the epilogue idle behind them.)
    const int npairs = nq * lq;
    int any = (pmask == nullptr) ? 1 : 0;
    if (pmask != nullptr)
        for (int i = tid; i < lp; i += BW_THREADS) any |= pmask[(int64_t)page * lp + i];
    int a_loc[PER_THREAD], rank_loc[PER_THREAD];
    float w_loc[PER_THREAD];
    auto load_pairs = [&](int c0) {
#pragma unroll
        for (int k = 0; k < PER_THREAD; ++k) {
            const int i = c0 + k * BW_THREADS + tid;
            int a = -1;
            float w = 0.f;
            if (i < npairs) {
                const int q = i / lq, n = i - q * lq;
                a = (int)argmax[((int64_t)q * np + page) * lq + n] - r0;
                w = g[(int64_t)q * np + page];
                if (qmask != nullptr && qmask[i] == 0) w = 0.f;
                if (a < 0 || a >= rows || w == 0.f) a = -1;
            }
            a_loc[k] = a;
            w_loc[k] = w;
        }
    };
    load_pairs(0);
    if constexpr (FUSED && EVDR_BW_EARLY_ROW == 2) {
        // ... and the epilogue's first PARAMETER row with them (9 registers carried through the bucketing; the moments, 16 more,
        // follow when the gather's registers are free): the workgroup's HBM stream starts with its first instruction
        if (gid < rows) load_row_x(nxt, gid);
    }
    if (tid == 0) sh_has = 0;
    for (int i = tid; i < rows * (EVDR_D / 4); i += BW_THREADS) reinterpret_cast<f32x4*>(acc)[i] = f32x4{0, 0, 0, 0};
    __syncthreads();
    if (any) sh_has = 1;                                  // benign race: every writer stores 1
    __syncthreads();
    auto bytesum = [](uint32_t x) { return (int)((x & 0xFFu) + ((x >> 8) & 0xFFu) + ((x >> 16) & 0xFFu) + (x >> 24)); };
    if (sh_has) {
        for (int c0 = 0; c0 < npairs; c0 += CHUNK) {
            for (int r = tid; r < BW_ROWS * CW; r += BW_THREADS) cnt[r] = 0u;
            if (tid == 0) sh_rounds = 0;
            if (c0 > 0) load_pairs(c0);
            __syncthreads();
            // (1) this thread's pairs: target row inside the slab (or -1), weight g * qmask, and the pair's RANK among the
            // lanes of this wave-instruction that hit the same row; per (row, instruction) counts
#pragma unroll
            for (int k = 0; k < PER_THREAD; ++k) {
                const int a = a_loc[k];
                // lanes of this instruction with the same row: AND over the 7 row bits of (bit set ? ballot : ~ballot)
                unsigned long long same = __ballot(a >= 0);
#pragma unroll
                for (int b = 0; b < 7; ++b) {
                    const bool bit = (a >> b) & 1;
                    const unsigned long long bb = __ballot(bit);
                    same &= bit ? bb : ~bb;
                }
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u));
                rank_loc[k] = rank;
                if (a >= 0 && rank == 0) {                              // one lane per row: integer add, order-free
                    const int slot = k * NWAVES + wave;
                    atomicAdd(&cnt[a * CW + (slot >> 2)], (uint32_t)__popcll(same) << (8 * (slot & 3)));
                }
            }
            __syncthreads();
            // (2) bucket sizes and their exclusive scan: BW_ROWS <= 128 buckets = the first one or two waves, one bucket per lane --
            // inclusive scan inside the wave by lane shifts, the second wave adds the first one's total: two barriers (the
            // Hillis-Steele form over all 512 threads took fifteen)
            int v = 0;
            if (tid < BW_ROWS) {
#pragma unroll
                for (int wd = 0; wd < CW; ++wd) v += bytesum(cnt[tid * CW + wd]);
                hist[tid] = v;
            }
            int incl = v;
            if (tid < BW_ROWS) {                           // whole waves: BW_ROWS is 64 or 128
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(incl, o);
                    if (lane >= o) incl += t;
                }
                if (lane == 63) sh_wsum[wave] = incl;
            }
            __syncthreads();
            if (tid < BW_ROWS) {
                const int base = (wave == 1) ? sh_wsum[0] : 0;
                offs[tid] = base + incl - v;               // exclusive
                if (tid == BW_ROWS - 1) sh_total = base + incl;    // all bucketed pairs of this chunk
            }
            __syncthreads();
            // (3) scatter: bucket start + pairs of the row in earlier instructions + rank inside this one = ascending pair order
#pragma unroll
            for (int k = 0; k < PER_THREAD; ++k) {
                const int a = a_loc[k];
                if (a >= 0) {
                    const int slot = k * NWAVES + wave;                 // wave-uniform
                    int before = 0;
#pragma unroll
                    for (int wd = 0; wd < CW; ++wd) {
                        const uint32_t c = cnt[a * CW + wd];
                        if (wd < (slot >> 2)) before += bytesum(c);
                        else if (wd == (slot >> 2)) before += bytesum(c & ((1u << (8 * (slot & 3))) - 1u));
                    }
                    const int pos = offs[a] + before + rank_loc[k];
                    list[pos] = (k * BW_THREADS + tid) | (a << 16);        // pair inside the chunk | target row
                    wl[pos] = w_loc[k];
                }
            }
            __syncthreads();
            // (4) the bucketed list (pairs sorted by target row) is cut into NGROUPS equal slices, one per 16-lane group,
            // whatever the distribution over rows: a patch that wins half of a page's (query, token) pairs -- salient
            // patches do -- no longer serialises on one group (all pairs on one row: 220 us before, now the uniform-case
            // time).  A group accumulates a row in registers with 4 query-row loads in flight and adds it into the slab when
            // the row changes: plain read-modify-write if the whole bucket lies in its slice.  The partial sum of a heavy row
            // it shares with its neighbours stays in registers -- `head`: a row that began in an earlier group's slice,
            // `tail`: a row that continues into the next group's -- and joins the slab in the ordered rounds below.
            if constexpr (FUSED && EVDR_BW_EARLY_ROW == 1) {
                // the epilogue's first parameter row is requested HERE, in front of the last chunk's gather: the HBM stream of the
                // workgroup starts a gather (four L2 round trips) earlier, and the row is in registers when the epilogue begins
                if (c0 + CHUNK >= npairs && gid < rows) load_row_x(nxt, gid);
            }
            {
                const int total = sh_total;
                const int per = ((total + NGROUPS - 1) / NGROUPS + 3) & ~3;
                // slice boundary g: g * per, moved back to the start of the bucket it falls into unless that bucket is
                // larger than a slice -- so only rows heavier than a slice are ever shared
                auto boundary = [&](int gi) {
                    const int x = gi * per;
                    if (x <= 0) return 0;
                    if (x >= total) return total;
                    const int r = list[x] >> 16;
                    return (hist[r] <= per) ? offs[r] : x;
                };
                const int lo = boundary(gid), hi = boundary(gid + 1);
                int cur_row = -1;
                f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
                int head_row = -1, head_turn = 0, tail_row = -1;
                f32x4 h0 = {0, 0, 0, 0}, h1 = {0, 0, 0, 0}, t0 = {0, 0, 0, 0}, t1 = {0, 0, 0, 0};
                auto flush = [&]() {
                    if (cur_row < 0) return;
                    const int b0 = offs[cur_row];
                    if (b0 >= lo && b0 + hist[cur_row] <= hi) {
                        float* dst = acc + cur_row * EVDR_D + sub * 8;
                        reinterpret_cast<f32x4*>(dst)[0] += s0;
                        reinterpret_cast<f32x4*>(dst)[1] += s1;
                    } else if (b0 < lo) {               // began in an earlier slice (only ever the first row of this one): this group
                        head_row = cur_row;             // is the row's (gid - first group)-th contributor; the first group of a heavy
                        head_turn = gid - b0 / per;     // row is the one whose slice holds its bucket start: floor(b0 / per)
                        h0 = s0;
                        h1 = s1;
                    } else {                            // begins here, continues in the next slice: the row's first contributor
                        tail_row = cur_row;
                        t0 = s0;
                        t1 = s1;
                    }
                };
                for (int k = lo; k < hi; k += 4) {
                    f32x4 v0[4], v1[4];
                    float w4[4];
                    int row4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int kk = min(k + u, hi - 1);
                        const int e = list[kk];
                        const float* qrow = Q + (int64_t)(c0 + (e & 0xFFFF)) * EVDR_D + sub * 8;
                        row4[u] = e >> 16;
                        w4[u] = wl[kk];
                        v0[u] = *reinterpret_cast<const f32x4*>(qrow);
                        v1[u] = *reinterpret_cast<const f32x4*>(qrow + 4);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (k + u < hi) {
                            if (row4[u] != cur_row) {
                                flush();
                                cur_row = row4[u];
                                s0 = f32x4{0, 0, 0, 0};
                                s1 = f32x4{0, 0, 0, 0};
                            }
                            s0 += v0[u] * w4[u];
                            s1 += v1[u] * w4[u];
                        }
                    }
                }
                flush();
                // ordered rounds over the shared rows: round 0 = every row's first contributor (tails: distinct rows), round
                // t = its t-th (heads with turn t: one group per row and round), plain read-modify-write, a barrier between
                // rounds.  The number of rounds is the longest chain of groups over one row (0 when nothing is shared).
                const int need = max(tail_row >= 0 ? 1 : 0, head_row >= 0 ? head_turn + 1 : 0);
                if (need > 0 && sub == 0) atomicMax(&sh_rounds, need);
                __syncthreads();
                const int rounds = sh_rounds;
                for (int t = 0; t < rounds; ++t) {
                    if (t == 0 && tail_row >= 0) {
                        float* dst = acc + tail_row * EVDR_D + sub * 8;
                        reinterpret_cast<f32x4*>(dst)[0] += t0;
                        reinterpret_cast<f32x4*>(dst)[1] += t1;
                    }
                    if (t > 0 && head_row >= 0 && head_turn == t) {
                        float* dst = acc + head_row * EVDR_D + sub * 8;
                        reinterpret_cast<f32x4*>(dst)[0] += h0;
                        reinterpret_cast<f32x4*>(dst)[1] += h1;
                    }
                    __syncthreads();
                }
            }
        }
    }
    if constexpr (!FUSED) {
        f32x4* out = reinterpret_cast<f32x4*>(dP + ((int64_t)page * lp + r0) * EVDR_D);
        for (int i = tid; i < rows * (EVDR_D / 4); i += BW_THREADS) out[i] = reinterpret_cast<const f32x4*>(acc)[i];
    } else {
        // one 16-lane group per parameter row: 8 floats per lane.  The next row's parameter and moments are requested before
        // this row's arithmetic and stores (two rows of loads in flight per group: the epilogue is a pure stream and each
        // group has only four rows to hide its latency behind)
        if (gid < rows) {
            if (EVDR_BW_EARLY_ROW) {
                // the early request of the parameter row sits inside the gather loop (EARLY_ROW == 1): a workgroup that never
                // enters it -- a page without a valid patch (sh_has == 0), or a zero-gradient step with no pairs at all -- loads the
                // row here, otherwise x would be the zeros of `RowIn nxt{}` and the decay / update would overwrite the parameter
                // (workgroup-uniform condition: no register carried through the gather)
                const bool have_x = (EVDR_BW_EARLY_ROW == 2) || (sh_has != 0 && npairs > 0);
                if (!have_x) load_row_x(nxt, gid);
                load_row_moments(nxt, gid);
            } else {
                nxt = load_row(gid);
            }
        }
        for (int r = gid; r < rows; r += NGROUPS) {
            const RowIn in = nxt;
            if (r + NGROUPS < rows) nxt = load_row(r + NGROUPS);
            const int64_t off = ((int64_t)page * lp + r0 + r) * EVDR_D + sub * 8;
            const float m = in.m;
            f32x4 x0 = in.x0, x1 = in.x1;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(acc + r * EVDR_D + sub * 8);
            const f32x4 g1 = *reinterpret_cast<const f32x4*>(acc + r * EVDR_D + sub * 8 + 4);
            // ---- backward of y = (m x) / (||m x|| + eps_n)
            const f32x4 v0 = x0 * m, v1 = x1 * m;
            float ss = v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2] + v0[3] * v0[3] + v1[0] * v1[0] + v1[1] * v1[1] +
                       v1[2] * v1[2] + v1[3] * v1[3];
            float dot = v0[0] * g0[0] + v0[1] * g0[1] + v0[2] * g0[2] + v0[3] * g0[3] + v1[0] * g1[0] + v1[1] * g1[1] +
                        v1[2] * g1[2] + v1[3] * g1[3];
            for (int o = 8; o > 0; o >>= 1) { ss += __shfl_xor(ss, o); dot += __shfl_xor(dot, o); }
            const float n = sqrtf(ss);
            const float inv = 1.f / (n + ad.eps_norm);
            const float c = (n > 0.f) ? dot * inv * inv / n : 0.f;
            const f32x4 d0 = (g0 * inv - v0 * c) * m, d1 = (g1 * inv - v1 * c) * m;
            // ---- AdamW (torch.optim.AdamW, amsgrad=False, maximize=False)
            f32x4 ea0 = in.ea0, ea1 = in.ea1;
            f32x4 es0 = in.es0, es1 = in.es1;
            const float decay = ad.decay;
            const float bc1 = ad.bc_dev ? ad.bc_dev[0] : ad.bc1;
            const float bc2_sqrt = ad.bc_dev ? ad.bc_dev[1] : ad.bc2_sqrt;
            const float step_size = ad.lr / bc1;
            x0 *= decay;
            x1 *= decay;
            ea0 = ea0 + (d0 - ea0) * ad.w1;                  // torch's lerp form of the first moment
            ea1 = ea1 + (d1 - ea1) * ad.w1;
            es0 = es0 * ad.beta2 + d0 * d0 * ad.w2;
            es1 = es1 * ad.beta2 + d1 * d1 * ad.w2;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                x0[k] -= step_size * (ea0[k] / (sqrtf(es0[k]) / bc2_sqrt + ad.eps));
                x1[k] -= step_size * (ea1[k] / (sqrtf(es1[k]) / bc2_sqrt + ad.eps));
            }
            *reinterpret_cast<f32x4*>(ad.x + off) = x0;
            *reinterpret_cast<f32x4*>(ad.x + off + 4) = x1;
            *reinterpret_cast<f32x4*>(ad.exp_avg + off) = ea0;
            *reinterpret_cast<f32x4*>(ad.exp_avg + off + 4) = ea1;
            *reinterpret_cast<f32x4*>(ad.exp_avg_sq + off) = es0;
            *reinterpret_cast<f32x4*>(ad.exp_avg_sq + off + 4) = es1;
            if (ad.next_hi != nullptr) {
                // ---- the next step's l2_normalize(m * x): same expressions, same 16-lane reduction as l2norm_fwd_kernel
                typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
                if (ad.pageflags != nullptr && m != 0.f) {
                    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
                    const u32x4 b0 = __builtin_bit_cast(u32x4, x0), b1 = __builtin_bit_cast(u32x4, x1);
                    bool bad = false;
#pragma unroll
                    for (int k = 0; k < 4; ++k) bad |= ((b0[k] & 0x7F800000u) == 0x7F800000u) || ((b1[k] & 0x7F800000u) == 0x7F800000u);
                    if (bad) atomicOr(&ad.pageflags[page], 8u);
                }
                f32x4 y0 = x0 * m, y1 = x1 * m;
                float s2 = y0[0] * y0[0] + y0[1] * y0[1] + y0[2] * y0[2] + y0[3] * y0[3] + y1[0] * y1[0] + y1[1] * y1[1] +
                           y1[2] * y1[2] + y1[3] * y1[3];
                for (int o = 8; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
                const float inv2 = 1.f / (sqrtf(s2) + ad.eps_norm);
                y0 *= inv2;
                y1 *= inv2;
                constexpr int K = 141 - 127;              // evdr_h2_shift(bits of 1.0f): |y| <= 1
                f16x8 a, b;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = __builtin_ldexpf((j < 4) ? y0[j & 3] : y1[j & 3], K);
                    const _Float16 h = (_Float16)f;
                    a[j] = h;
                    b[j] = (_Float16)(f - (float)h);
                }
                *reinterpret_cast<f16x8*>(ad.next_hi + off) = a;
                *reinterpret_cast<f16x8*>(ad.next_lo + off) = b;
            }
        }
        if (ad.next_amax != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *ad.next_amax = 0x3F800000u;
    }
    (void)lane;
}

// ---- gradient w.r.t. the QUERY embeddings:  dQ[q,n,:] = qmask[q,n] * sum_p g[q,p] * has(p) * P[p, argmax[q,p,n], :]
// One 16-lane group per (query, token); the np (argmax, weight) pairs are fetched 16 at a time, one per lane, and the
// page rows (512 B, fp32) are gathered four at a time in registers.  HBM/L2-bound: nq*lq*np rows of 512 B.
__global__ void __launch_bounds__(256) maxsim_bwd_q_kernel(const float* __restrict__ g, const float* __restrict__ P,
                                                          const uint8_t* __restrict__ qmask,
                                                          const uint32_t* __restrict__ pageflags,
                                                          const uint16_t* __restrict__ argmax, float* __restrict__ dQ,
                                                          int nq, int lq, int np, int lp, int pages_per_seg) {
    // blockIdx.y = page segment: with few (query, token) pairs (a training batch has 1024: 64 workgroups for 256 CUs) the
    // page range is cut so that the grid fills the chip; each segment writes its partial sums to its own slab of the
    // workspace and reduce_segments_kernel adds the slabs in segment order: dQ is the same bits run after run (global
    // float atomics, the first version, were not)
    const int sub = threadIdx.x & 15;
    const int pair = blockIdx.x * 16 + (threadIdx.x >> 4);          // (q, n) handled by this 16-lane group
    if (pair >= nq * lq) return;
    const int q = pair / lq, n = pair - q * lq;
    const int pbeg = blockIdx.y * pages_per_seg, pend = min(np, pbeg + pages_per_seg);
    f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
    const bool live = (qmask == nullptr) || qmask[pair] != 0;
    if (live) {
        for (int p0 = pbeg; p0 < pend; p0 += 16) {
            const int pl = p0 + sub;                                // this lane's page of the batch
            int a = 0;
            float w = 0.f;
            if (pl < pend) {
                w = (pageflags[pl] & 1u) ? g[(int64_t)q * np + pl] : 0.f;       // has(p) = page has a valid patch
                a = (int)argmax[((int64_t)q * np + pl) * lq + n];
            }
#pragma unroll
            for (int k0 = 0; k0 < 16; k0 += 4) {
                f32x4 v0[4], v1[4];
                float wk[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int src = (threadIdx.x & 48) + k0 + u;    // lane k0+u of this 16-lane group (wave-relative)
                    const int ak = __shfl(a, src);
                    wk[u] = __shfl(w, src);
                    const int pk = min(p0 + k0 + u, np - 1);
                    const float* row = P + ((int64_t)pk * lp + ak) * EVDR_D + sub * 8;
                    v0[u] = *reinterpret_cast<const f32x4*>(row);
                    v1[u] = *reinterpret_cast<const f32x4*>(row + 4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { s0 += v0[u] * wk[u]; s1 += v1[u] * wk[u]; }
            }
        }
    }
    float* dst = dQ + ((int64_t)blockIdx.y * nq * lq + pair) * EVDR_D + sub * 8;     // dQ = the partial slabs when gridDim.y > 1
    *reinterpret_cast<f32x4*>(dst) = s0;
    *reinterpret_cast<f32x4*>(dst + 4) = s1;
}

// out[i] = partials[0][i] + partials[1][i] + ... in that order (n4 float4 per slab)
__global__ void __launch_bounds__(256) reduce_segments_kernel(const float* __restrict__ partials, int nseg, int64_t n4,
                                                             float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 acc = reinterpret_cast<const f32x4*>(partials)[i];
        for (int sgm = 1; sgm < nseg; ++sgm) acc += reinterpret_cast<const f32x4*>(partials)[(int64_t)sgm * n4 + i];
        reinterpret_cast<f32x4*>(out)[i] = acc;
    }
}

// ---- l2_normalize (utils/preprocess_data.py:8-9) with an optional per-row mask, forward and backward ---------------
// y = m * x / (||m*x|| + eps);  one 16-lane group per 128-wide row (8 floats per lane, two 16-B accesses).
// backward of y = x/(n + eps):  dx = g/(n+eps) - x * (x.g) / (n (n+eps)^2)   (n > 0; the norm's subgradient at 0 is 0)
// SPLIT: also (or only, y == null) emit y as the fp16 hi/lo planes the forward kernel reads (evdr_split_f32's format).
// |y| <= 1 by construction, so the scale is the constant 2^14 and the absmax word is that of 1.0f: no absmax pass and no
// second trip of the normalised pages through HBM on the training step.
template <bool SPLIT>
__global__ void __launch_bounds__(256) l2norm_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ rowmask,
                                                        int64_t rows, float eps, float* __restrict__ y,
                                                        float* __restrict__ norm, _Float16* __restrict__ hi,
                                                        _Float16* __restrict__ lo, uint32_t* __restrict__ amax_bits,
                                                        uint32_t* __restrict__ pageflags, int64_t rows_per_page) {
    typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
    const int sub = threadIdx.x & 15;
    if (SPLIT && blockIdx.x == 0 && threadIdx.x == 0) *amax_bits = 0x3F800000u;
    for (int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); r < rows; r += (int64_t)gridDim.x * 16) {
        const float m = (rowmask == nullptr || rowmask[r] != 0) ? 1.f : 0.f;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(x + r * EVDR_D + sub * 8);
        f32x4 v1 = *reinterpret_cast<const f32x4*>(x + r * EVDR_D + sub * 8 + 4);
        if (pageflags != nullptr && m != 0.f) {
            // a NaN / Inf element of an unmasked row (tested on the LOADED bits: the build lets the optimiser assume that
            // float arithmetic never yields NaN): the normalised row is NaN as in the reference's x / (norm + eps), and the
            // page is reported so that the scorer returns NaN for it (evdr.h, "non-finite inputs")
            typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
            const u32x4 b0 = __builtin_bit_cast(u32x4, v0), b1 = __builtin_bit_cast(u32x4, v1);
            bool bad = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) bad |= ((b0[k] & 0x7F800000u) == 0x7F800000u) || ((b1[k] & 0x7F800000u) == 0x7F800000u);
            if (bad) atomicOr(&pageflags[r / rows_per_page], 8u);
        }
        v0 *= m;
        v1 *= m;
        float ss = v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2] + v0[3] * v0[3] + v1[0] * v1[0] + v1[1] * v1[1] +
                   v1[2] * v1[2] + v1[3] * v1[3];
        for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        const float n = sqrtf(ss);
        const float inv = 1.f / (n + eps);
        v0 *= inv;
        v1 *= inv;
        if (y != nullptr) {
            *reinterpret_cast<f32x4*>(y + r * EVDR_D + sub * 8) = v0;
            *reinterpret_cast<f32x4*>(y + r * EVDR_D + sub * 8 + 4) = v1;
        }
        if (sub == 0 && norm != nullptr) norm[r] = n;
        if constexpr (SPLIT) {
            constexpr int K = 141 - 127;              // evdr_h2_shift(bits of 1.0f)
            f16x8 a, b;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = __builtin_ldexpf((j < 4) ? v0[j & 3] : v1[j & 3], K);
                const _Float16 h = (_Float16)f;
                a[j] = h;
                b[j] = (_Float16)(f - (float)h);
            }
            *reinterpret_cast<f16x8*>(hi + r * EVDR_D + sub * 8) = a;
            *reinterpret_cast<f16x8*>(lo + r * EVDR_D + sub * 8) = b;
        }
    }
}

__global__ void __launch_bounds__(256) l2norm_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                        const uint8_t* __restrict__ rowmask, const float* __restrict__ norm,
                                                        int64_t rows, float eps, float* __restrict__ dx) {
    const int sub = threadIdx.x & 15;
    for (int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); r < rows; r += (int64_t)gridDim.x * 16) {
        const float m = (rowmask == nullptr || rowmask[r] != 0) ? 1.f : 0.f;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(x + r * EVDR_D + sub * 8);
        f32x4 v1 = *reinterpret_cast<const f32x4*>(x + r * EVDR_D + sub * 8 + 4);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gy + r * EVDR_D + sub * 8);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(gy + r * EVDR_D + sub * 8 + 4);
        v0 *= m;
        v1 *= m;
        float dot = v0[0] * g0[0] + v0[1] * g0[1] + v0[2] * g0[2] + v0[3] * g0[3] + v1[0] * g1[0] + v1[1] * g1[1] +
                    v1[2] * g1[2] + v1[3] * g1[3];
        for (int o = 8; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
        const float n = norm[r];
        const float inv = 1.f / (n + eps);
        const float c = (n > 0.f) ? dot * inv * inv / n : 0.f;
        *reinterpret_cast<f32x4*>(dx + r * EVDR_D + sub * 8) = (g0 * inv - v0 * c) * m;
        *reinterpret_cast<f32x4*>(dx + r * EVDR_D + sub * 8 + 4) = (g1 * inv - v1 * c) * m;
    }
}

// ---- infonce_distillation_loss (criterion.py:56-68) and d loss / d score_s, one workgroup per query row
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

// ticket != null: the workgroup that takes the last ticket also reduces the row losses (mean_kernel's order) -- one launch
template <bool TICKET>
__global__ void __launch_bounds__(256) infonce_row_kernel(const float* __restrict__ ss, const float* __restrict__ st,
                                                         int64_t n, float temp, float inv_tb,
                                                         float* __restrict__ row_loss, float* __restrict__ dscore,
                                                         unsigned int* __restrict__ ticket, float* __restrict__ loss) {
    __shared__ float red[4];
    __shared__ uint32_t redk[4];
    __shared__ int redi[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* s = ss + (int64_t)blockIdx.x * n;
    const float* t = st + (int64_t)blockIdx.x * n;
    // teacher argmax as torch.argmax defines it (criterion.py:61): first maximal index, -0.0 == +0.0, and a NaN counts as
    // the greatest value (first NaN wins) -- on order keys, so that an all -inf or all-NaN row still yields a VALID index
    // (0 / the first NaN) instead of leaving a sentinel that would be used as an address below
    uint32_t tbest = 0u;
    float smax = -__builtin_inff();
    int tidx = 0;
    // the row is read from memory ONCE: a thread's first KEEP elements of s / temp stay in registers for the two later passes (all
    // of them for n <= 1024: a training batch scores 500 pages); the same values in the same order as re-reading them
    constexpr int KEEP = 4;
    float z[KEEP];
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        const int64_t i = tid + 256 * k;
        z[k] = (i < n) ? s[i] / temp : -__builtin_inff();
    }
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        const int64_t i = tid + 256 * k;
        if (i < n) {
            const uint32_t tk = nan_max_key(t[i]);
            if (tk > tbest) { tbest = tk; tidx = (int)i; }
            smax = fmaxf(smax, z[k]);
        }
    }
    for (int64_t i = tid + 256 * KEEP; i < n; i += 256) {
        const uint32_t tk = nan_max_key(t[i]);
        if (tk > tbest) { tbest = tk; tidx = (int)i; }
        smax = fmaxf(smax, s[i] / temp);
    }
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t ov = (uint32_t)__shfl_xor((int)tbest, o);
        const int oi = __shfl_xor(tidx, o);
        if (ov > tbest || (ov == tbest && oi < tidx)) { tbest = ov; tidx = oi; }
    }
    smax = wave_max(smax);
    if (lane == 0) { redk[wave] = tbest; redi[wave] = tidx; }
    __syncthreads();
    tbest = redk[0]; tidx = redi[0];
    for (int w = 1; w < 4; ++w)
        if (redk[w] > tbest || (redk[w] == tbest && redi[w] < tidx)) { tbest = redk[w]; tidx = redi[w]; }
    if (tidx < 0 || tidx >= n) tidx = 0;          // threads that saw no element carry index 0 with key 0: never out of range
    __syncthreads();
    if (lane == 0) red[wave] = smax;
    __syncthreads();
    smax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    float e[KEEP];
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        e[k] = expf(z[k] - smax);
        if (tid + 256 * k < n) sum += e[k];
    }
    for (int64_t i = tid + 256 * KEEP; i < n; i += 256) sum += expf(s[i] / temp - smax);
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = red[0] + red[1] + red[2] + red[3];
    const float lse = smax + logf(sum);
    if (tid == 0) row_loss[blockIdx.x] = lse - s[tidx] / temp;
    if (dscore != nullptr) {
        float* d = dscore + (int64_t)blockIdx.x * n;
        const float inv_sum = 1.f / sum;
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int64_t i = tid + 256 * k;
            if (i < n) d[i] = (e[k] * inv_sum - ((int)i == tidx ? 1.f : 0.f)) * inv_tb;
        }
        for (int64_t i = tid + 256 * KEEP; i < n; i += 256) {
            const float pr = expf(s[i] / temp - smax) * inv_sum;
            d[i] = (pr - ((int)i == tidx ? 1.f : 0.f)) * inv_tb;
        }
    }
    if constexpr (TICKET) {
        __shared__ int is_last;
        __syncthreads();                                   // tid 0's row_loss store is ordered before its fence below
        if (tid == 0) {
            __threadfence();
            is_last = (atomicAdd(ticket, 1u) == gridDim.x - 1) ? 1 : 0;
        }
        __syncthreads();
        if (is_last) {
            __threadfence();
            const volatile float* rl = row_loss;            // written by other CUs: read past this CU's vector L1
            float v = 0.f;
            for (int64_t i = tid; i < gridDim.x; i += 256) v += rl[i];
            v = wave_sum(v);
            __syncthreads();
            if (lane == 0) red[wave] = v;
            __syncthreads();
            if (tid == 0) {
                loss[0] = (red[0] + red[1] + red[2] + red[3]) / (float)gridDim.x;
                *ticket = 0u;                               // ready for the next call
            }
        }
    }
}

__global__ void __launch_bounds__(256) mean_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) / (float)n;
}

}  // namespace

template <int BW_ROWS, int CHUNK, bool FUSED>
static hipError_t launch_bwd(const float* g, const float* Q, const uint8_t* qmask, const uint8_t* pmask, const uint16_t* argmax,
                             float* dP, int64_t nq, int64_t lq, int64_t np, int64_t lp, const AdamArgs& ad, hipStream_t stream) {
    constexpr int CW = (CHUNK / BW_THREADS * (BW_THREADS / 64) + 3) / 4;
    constexpr int LDS = BW_ROWS * EVDR_D * 4 + CHUNK * 8 + BW_ROWS * 8 + BW_ROWS * CW * 4;
    auto kern = maxsim_bwd_kernel<BW_ROWS, CHUNK, FUSED>;
    static std::atomic<uint64_t> attr_devs{0};
    if (hipError_t e = evdr_ensure_dyn_lds((const void*)kern, LDS, attr_devs); e != hipSuccess) return e;
    const int64_t nslabs = (lp + BW_ROWS - 1) / BW_ROWS;
    const int slab_rows = (int)((lp + nslabs - 1) / nslabs);            // equally tall slabs, each within the capacity
    dim3 grid((unsigned)np, (unsigned)nslabs);
    hipLaunchKernelGGL(kern, grid, dim3(BW_THREADS), LDS, stream, g, Q, qmask, pmask, argmax, dP, (int)nq, (int)lq, (int)np,
                       (int)lp, slab_rows, ad);
    return hipGetLastError();
}

template <bool FUSED>
static hipError_t dispatch_bwd(const float* g, const float* Q, const uint8_t* qmask, const uint8_t* pmask, const uint16_t* argmax,
                               float* dP, int64_t nq, int64_t lq, int64_t np, int64_t lp, const AdamArgs& ad, hipStream_t stream) {
    if (np == 0 || lp == 0) return hipSuccess;
    // 128-row slabs (64 KiB of LDS + lists: two workgroups per CU), also for compressed pages of <= 256 patches that would fit
    // one 256-row slab: with one workgroup per CU the CU's HBM stream stops while that workgroup buckets and gathers, with
    // two the parameter / moment traffic of one overlaps the gather of the other -- 72 -> 62 us for the fused update at
    // B = 32, N = 500, Ls = 206 (316 MB of x / exp_avg / exp_avg_sq traffic: 5.1 TB/s), although every pair is bucketed twice
    // Few workgroups (a page shard of a multi-GPU run: 63 pages x 2 slabs = 126 workgroups on 256 CUs): 64-row slabs (41 KiB of
    // LDS: three workgroups per CU, twice the workgroups) -- 25.8 -> 17.7 us at 63 pages x 206 patches; with >= one workgroup
    // per slot of the 128-row form the extra bucketing passes cost more than they hide (81.8 against 73.6 us at 500 pages).
#if defined(EVDR_BW_FORCE_CAP) && EVDR_BW_FORCE_CAP == 64
    return launch_bwd<64, 1024, FUSED>(g, Q, qmask, pmask, argmax, dP, nq, lq, np, lp, ad, stream);
#elif defined(EVDR_BW_FORCE_CAP) && EVDR_BW_FORCE_CAP == 128
    return launch_bwd<128, 1024, FUSED>(g, Q, qmask, pmask, argmax, dP, nq, lq, np, lp, ad, stream);
#else
    if (lp > 64 && np * ((lp + 127) / 128) <= EVDR_BW_SMALL_WGS) return launch_bwd<64, 1024, FUSED>(g, Q, qmask, pmask, argmax, dP, nq, lq, np, lp, ad, stream);
    return launch_bwd<128, 1024, FUSED>(g, Q, qmask, pmask, argmax, dP, nq, lq, np, lp, ad, stream);
#endif
}

hipError_t evdr_launch_maxsim_bwd(const float* g, const float* Q, const uint8_t* qmask, const uint8_t* pmask,
                                  const uint16_t* argmax, float* dP, int64_t nq, int64_t lq, int64_t np, int64_t lp,
                                  hipStream_t stream) {
    return dispatch_bwd<false>(g, Q, qmask, pmask, argmax, dP, nq, lq, np, lp, AdamArgs{}, stream);
}

hipError_t evdr_launch_maxsim_bwd_adamw(const float* g, const float* Q, const uint8_t* qmask, const uint8_t* pmask,
                                        const uint16_t* argmax, float* x, float* exp_avg, float* exp_avg_sq, int64_t nq,
                                        int64_t lq, int64_t np, int64_t lp, double lr, double beta1, double beta2, double eps,
                                        double weight_decay, double bc1, double bc2_sqrt, float eps_norm, const void* state,
                                        void* next_planes, uint32_t* next_amax, uint32_t* pageflags, hipStream_t stream) {
    AdamArgs ad{x, exp_avg, exp_avg_sq, (float)lr, (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2,
                (float)(1.0 - beta2), (float)eps, (float)bc1, (float)bc2_sqrt, eps_norm,
                state ? reinterpret_cast<const float*>(reinterpret_cast<const long long*>(state) + 1) : nullptr,
                (_Float16*)next_planes, next_planes ? (_Float16*)next_planes + np * lp * EVDR_D : nullptr, next_amax, pageflags};
    return dispatch_bwd<true>(g, Q, qmask, pmask, argmax, nullptr, nq, lq, np, lp, ad, stream);
}

hipError_t evdr_launch_adamw(const float* g, float* x, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                             double beta2, double eps, double weight_decay, double bc1, double bc2_sqrt, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const int64_t n4 = (n + 3) / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;                     // grid-stride: 16 workgroups per CU keep the stream full
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, g, x, exp_avg, exp_avg_sq, n,
                       (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                       (float)(lr / bc1), (float)bc2_sqrt, (float)eps);
    return hipGetLastError();
}

hipError_t evdr_launch_adamw_advance(void* state, double beta1, double beta2, hipStream_t stream) {
    hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(1), 0, stream, state, beta1, beta2);
    return hipGetLastError();
}

// page segments of the dQ kernel (1 = every (query, token) pair walks all pages itself)
static void bwd_q_geometry(int64_t pairs, int64_t np, int64_t& nseg, int64_t& per) {
    const int64_t bx = (pairs + 15) / 16;
    nseg = bx > 0 ? (1024 + bx - 1) / bx : 1;                       // ~1024 workgroups in total ...
    const int64_t max_seg = (np + 63) / 64;                         // ... of at least 64 pages each
    if (nseg > max_seg) nseg = max_seg;
    if (nseg < 1) nseg = 1;
    per = ((np + nseg - 1) / nseg + 15) / 16 * 16;
    nseg = per > 0 ? (np + per - 1) / per : 1;
    if (nseg < 1) nseg = 1;
    if (per < 16) per = 16;
}
int evdr_bwd_q_segments(int64_t pairs, int64_t np) {
    int64_t nseg, per;
    bwd_q_geometry(pairs, np, nseg, per);
    return nseg > 1 ? (int)nseg : 0;                                // slabs of workspace needed
}

hipError_t evdr_launch_maxsim_bwd_q(const float* g, const float* P, const uint8_t* qmask, const uint32_t* pageflags,
                                    const uint16_t* argmax, float* dQ, float* partials, int64_t nq, int64_t lq, int64_t np,
                                    int64_t lp, hipStream_t stream) {
    const int64_t pairs = nq * lq;
    if (pairs == 0) return hipSuccess;
    int64_t nseg, per;
    bwd_q_geometry(pairs, np, nseg, per);
    const int64_t bx = (pairs + 15) / 16;
    hipLaunchKernelGGL(maxsim_bwd_q_kernel, dim3((unsigned)bx, (unsigned)nseg), dim3(256), 0, stream, g, P, qmask, pageflags,
                       argmax, nseg > 1 ? partials : dQ, (int)nq, (int)lq, (int)np, (int)lp, (int)per);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || nseg == 1) return e;
    const int64_t n4 = pairs * (EVDR_D / 4);
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(reduce_segments_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)partials, (int)nseg, n4, dQ);
    return hipGetLastError();
}

hipError_t evdr_launch_l2norm_fwd(const float* x, const uint8_t* rowmask, int64_t rows, float eps, float* y, float* norm,
                                  uint16_t* planes, uint32_t* amax_bits, uint32_t* pageflags, int64_t rows_per_page,
                                  hipStream_t stream) {
    if (rows_per_page < 1) rows_per_page = 1;
    if (rows == 0) return hipSuccess;
    int64_t blocks = (rows + 15) / 16;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (planes != nullptr)
        hipLaunchKernelGGL(l2norm_fwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, x, rowmask, rows, eps, y, norm,
                           (_Float16*)planes, (_Float16*)planes + rows * EVDR_D, amax_bits, pageflags, rows_per_page);
    else
        hipLaunchKernelGGL(l2norm_fwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, x, rowmask, rows, eps, y, norm,
                           (_Float16*)nullptr, (_Float16*)nullptr, (uint32_t*)nullptr, pageflags, rows_per_page);
    return hipGetLastError();
}

hipError_t evdr_launch_l2norm_bwd(const float* gy, const float* x, const uint8_t* rowmask, const float* norm, int64_t rows,
                                  float eps, float* dx, hipStream_t stream) {
    if (rows == 0) return hipSuccess;
    int64_t blocks = (rows + 15) / 16;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, gy, x, rowmask, norm, rows, eps, dx);
    return hipGetLastError();
}

hipError_t evdr_launch_infonce_ws(const float* ss, const float* st, int64_t b, int64_t n, float temperature, float* loss,
                                  float* dscore, float* workspace, hipStream_t stream) {
    if (b == 0) return hipSuccess;
    hipLaunchKernelGGL(infonce_row_kernel<true>, dim3((unsigned)b), dim3(256), 0, stream, ss, st, n, temperature,
                       1.f / (temperature * (float)b), workspace, dscore, reinterpret_cast<unsigned int*>(workspace + b), loss);
    return hipGetLastError();
}

hipError_t evdr_launch_infonce(const float* ss, const float* st, int64_t b, int64_t n, float temperature, float* loss,
                               float* dscore, float* row_loss, hipStream_t stream) {
    if (b == 0) return hipSuccess;
    hipLaunchKernelGGL(infonce_row_kernel<false>, dim3((unsigned)b), dim3(256), 0, stream, ss, st, n, temperature,
                       1.f / (temperature * (float)b), row_loss, dscore, (unsigned int*)nullptr, (float*)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, stream, row_loss, b, loss);
    return hipGetLastError();
}
