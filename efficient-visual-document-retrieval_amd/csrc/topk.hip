// Per-query top-k over a (nq, n) fp32 score matrix, entirely on device.  Replaces the reference's
// Nq*N `.item()` loop that moves every score to the host before ranking
// (mainv2_iter_distill_infonce.py:311-317); k = 100 covers every cut-off in evaluator/retrieval.py:223.
//
// One 256-thread workgroup per row: 3 radix-select passes (11 + 11 + 10 bits, LDS histogram of 2048 bins) find the k-th
// largest key exactly, one pass gathers the candidates, a 128-wide bitonic sort orders them.  Order is total and
// deterministic: score descending, reported index ascending on ties (the shard merge relies on it).
// HBM-bound integer work: the row is read once from HBM, the re-reads hit L2 (400 KB per 100k-page row).
// The first digit (sign, exponent, two mantissa bits) is the same for almost every score of a row, and 64 LDS atomics on
// one address serialise: that pass counts per WAVE (ballot + popcount, one atomic per distinct digit), the others use
// plain atomics on digits that spread over the bins.
#include "evdr_common.h"

namespace {

constexpr int TK_THREADS = 256;
constexpr int TK_BINS = 2048;
constexpr int TK_BPT = TK_BINS / TK_THREADS;     // histogram bins per thread in the suffix scan

__device__ __forceinline__ uint32_t order_key(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0xFFFFFFFFu;   // NaN of either sign: ranks first, like torch.topk (integer test:
                                                                // the build assumes no NaNs in float compares)
    if (u == 0x80000000u) u = 0u;                    // -0.0 -> +0.0 so that equal scores tie
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __builtin_bit_cast(float, u);
}

// nseg > 1: workgroup b ranks segment b % nseg (columns [seg * (b % nseg), +seg) of row b / nseg) and writes that segment's
// k candidates; a second launch over the nseg * k candidates of each row (idx_map = their indices) finishes the job.  One
// workgroup per row leaves 255 CUs idle when a handful of queries rank 100 k pages.
__global__ void __launch_bounds__(TK_THREADS) topk_kernel(const float* __restrict__ scores,
                                                         const int32_t* __restrict__ idx_map, int64_t n_total,
                                                         int64_t row_stride, int32_t idx_base, int k, int nseg, int64_t seg,
                                                         float* __restrict__ top_scores,
                                                         int32_t* __restrict__ top_idx) {
    __shared__ uint32_t hist[TK_BINS];
    __shared__ uint32_t suf[TK_THREADS + 1];
    __shared__ unsigned long long cand[EVDR_TOPK_MAX];
    __shared__ uint32_t sh_prefix, sh_need, sh_cnt_eq, sh_cnt_gt_slot, sh_eq_slot, sh_wave_tot[TK_THREADS / 64];
    __shared__ uint32_t sh_eq_base;

    const int tid = threadIdx.x;
    const int64_t rowi = blockIdx.x / nseg;
    const int64_t col0 = (int64_t)(blockIdx.x % nseg) * seg;
    const int64_t n = max((int64_t)0, min(seg, n_total - col0));
    const float* row = scores + rowi * row_stride + col0;
    const int32_t* map = idx_map ? idx_map + rowi * n_total + col0 : nullptr;
    idx_base += (int32_t)col0;
    const int64_t orow = blockIdx.x;                 // output row: (row, segment)
    const int keff = (int)min((int64_t)k, n);
    // Row scan with 16-B loads, 16-32 elements in flight per thread (one workgroup has to keep enough bytes in flight to cover the
    // L2 latency: scalar loads left it at ~4 GB/s per row).  body(valid, key, index) is called the same number of times by
    // every thread, so it may use wave-level votes.
    const int mis = (int)((reinterpret_cast<uintptr_t>(row) >> 2) & 3);            // floats past a 16-B boundary
    const int head = (int)min((int64_t)((4 - mis) & 3), n);
    const int64_t nvec = (n - head) / 4;
    const int tail = (int)(n - head - 4 * nvec);
    const f32x4* r4 = reinterpret_cast<const f32x4*>(row + head);
    auto scan = [&](auto&& body) {
        {
            int64_t i = -1;
            if (tid < head) i = tid;
            else if (tid < head + tail) i = n - tail + (tid - head);
            const bool v = i >= 0;
            body(v, v ? order_key(row[i]) : 0u, i);
        }
        // software-pipelined: the loads of the next round are in flight while this round's keys are counted
        constexpr int U = 4;
        f32x4 cur[U], nxt[U];
        auto fetch = [&](f32x4 (&x)[U], int64_t v0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t j = v0 + u * TK_THREADS + tid;
                x[u] = (j < nvec) ? r4[j] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        fetch(cur, 0);
        for (int64_t v0 = 0; v0 < nvec; v0 += U * TK_THREADS) {
            if (v0 + U * TK_THREADS < nvec) fetch(nxt, v0 + U * TK_THREADS);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t j = v0 + u * TK_THREADS + tid;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float f = cur[u][e];
                    body(j < nvec, order_key(f), head + 4 * j + e);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        }
    };

    if (tid < EVDR_TOPK_MAX) cand[tid] = 0ull;       // key 0 sorts below every real score
    if (tid == 0) { sh_prefix = 0; sh_need = (uint32_t)keff; sh_cnt_gt_slot = 0; sh_eq_slot = 0; sh_eq_base = 0; }
    __syncthreads();

    if (keff > 0) {
        // ---- radix select, most significant digit first
        const int lane = tid & 63;
        for (int pass = 0; pass < 3; ++pass) {
            const int shift = (pass == 0) ? 21 : (pass == 1 ? 10 : 0);
            const int bits = (pass == 2) ? 10 : 11;
            const uint32_t dmask = (1u << bits) - 1u;
#pragma unroll
            for (int b = 0; b < TK_BPT; ++b) hist[tid * TK_BPT + b] = 0;
            __syncthreads();
            const uint32_t prefix = sh_prefix;
            const uint32_t himask = (pass == 0) ? 0u : (0xFFFFFFFFu << (shift + bits));
            scan([&](bool valid, uint32_t key, int64_t) {
                const uint32_t d = (key >> shift) & dmask;
                bool mine = valid && ((key & himask) == prefix);
                if (pass == 0) {
                    // wave-level counting of the (few) distinct digits; whatever is left after 3 rounds goes one by one
                    unsigned long long todo = __ballot(mine);
                    for (int it = 0; it < 3 && todo != 0ull; ++it) {
                        const int leader = __ffsll((long long)todo) - 1;
                        const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)d, leader);
                        const unsigned long long same = __ballot(mine && d == dl);
                        if (lane == leader) atomicAdd(&hist[dl], (uint32_t)__popcll(same));
                        todo &= ~same;
                        mine = mine && d != dl;
                    }
                }
                if (mine) atomicAdd(&hist[d], 1u);
            });
            __syncthreads();
            // suffix sums over the bins, TK_BPT bins per thread: suf[t] = count of digits >= TK_BPT * t
            uint32_t own = 0;
#pragma unroll
            for (int b = 0; b < TK_BPT; ++b) own += hist[tid * TK_BPT + b];
            suf[tid] = own;
            if (tid == 0) suf[TK_THREADS] = 0;
            __syncthreads();
            for (int off = 1; off < TK_THREADS; off <<= 1) {
                uint32_t add = (tid + off < TK_THREADS) ? suf[tid + off] : 0u;
                __syncthreads();
                suf[tid] += add;
                __syncthreads();
            }
            const uint32_t need = sh_need;
            if (suf[tid] >= need && suf[tid + 1] < need) {      // exactly one thread: the wanted digit is one of its bins
                uint32_t above = suf[tid + 1];
                for (int b = TK_BPT - 1; b >= 0; --b) {
                    const uint32_t cnt = hist[tid * TK_BPT + b];
                    if (above + cnt >= need) {
                        sh_prefix = prefix | ((uint32_t)(tid * TK_BPT + b) << shift);
                        sh_need = need - above;
                        sh_cnt_eq = cnt;
                        break;
                    }
                    above += cnt;
                }
            }
            __syncthreads();
        }
        const uint32_t T = sh_prefix;             // key of the k-th largest element
        const uint32_t need_eq = sh_need;         // how many elements == T belong to the top-k
        const uint32_t n_gt = (uint32_t)keff - need_eq;
        const bool take_all_eq = (sh_cnt_eq == need_eq);

        // ---- gather: everything above T, plus (all | the first need_eq in index order) of == T
        scan([&](bool valid, uint32_t key, int64_t i) {
            const bool gt = valid && key > T;
            const bool eq = valid && take_all_eq && key == T;
            if (gt || eq) {
                const uint32_t slot = gt ? atomicAdd(&sh_cnt_gt_slot, 1u) : n_gt + atomicAdd(&sh_eq_slot, 1u);
                const uint32_t rep = (uint32_t)(map ? map[i] : (int32_t)i + idx_base);
                cand[slot] = ((unsigned long long)key << 32) | (0xFFFFFFFFu - rep);
            }
        });
        if (!take_all_eq) {
            // the tie straddles the cut: ordered scan, stop once need_eq have been taken
            const int wv = tid >> 6;
            for (int64_t base = 0; base < n; base += TK_THREADS) {
                const int64_t i = base + tid;
                const bool eq = (i < n) && order_key(row[i]) == T;
                const unsigned long long bal = __ballot(eq);
                if (lane == 0) sh_wave_tot[wv] = (uint32_t)__popcll(bal);
                __syncthreads();
                uint32_t before = sh_eq_base;
                for (int w = 0; w < wv; ++w) before += sh_wave_tot[w];
                const uint32_t rank = before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                if (eq && rank < need_eq) {
                    const uint32_t rep = (uint32_t)(map ? map[i] : (int32_t)i + idx_base);
                    cand[n_gt + rank] = ((unsigned long long)T << 32) | (0xFFFFFFFFu - rep);
                }
                __syncthreads();
                if (tid == 0) {
                    uint32_t tot = 0;
                    for (int w = 0; w < TK_THREADS / 64; ++w) tot += sh_wave_tot[w];
                    sh_eq_base += tot;
                }
                __syncthreads();
                if (sh_eq_base >= need_eq) break;
            }
        }
    }
    __syncthreads();

    // ---- bitonic sort, descending, 128 slots (unused slots hold 0 and sink to the end)
    for (int size = 2; size <= EVDR_TOPK_MAX; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (tid < EVDR_TOPK_MAX / 2) {
                const int lo = 2 * tid - (tid & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = cand[lo], b = cand[hi];
                if ((a < b) == desc) { cand[lo] = b; cand[hi] = a; }
            }
            __syncthreads();
        }
    }
    if (tid < k) {
        const unsigned long long c = cand[tid];
        const bool real = tid < keff;
        top_scores[orow * k + tid] = real ? key_to_float((uint32_t)(c >> 32)) : -__builtin_inff();
        top_idx[orow * k + tid] = real ? (int32_t)(0xFFFFFFFFu - (uint32_t)c) : -1;
    }
}

}  // namespace

// Segments per row for the two-level form (1 = one workgroup per row is enough): aim at ~512 workgroups of >= 4096 columns.
int evdr_topk_segments(int64_t nq, int64_t n) {
    if (nq >= 256 || n < 2 * 4096) return 1;
    int64_t by_len = (n + 4095) / 4096, by_fill = 512 / (nq > 0 ? nq : 1);
    int64_t s = by_len < by_fill ? by_len : by_fill;
    return (int)(s < 1 ? 1 : s);
}

hipError_t evdr_launch_topk(const float* scores, const int32_t* idx_map, int64_t nq, int64_t n, int64_t row_stride,
                            int32_t idx_base, int k, float* top_scores, int32_t* top_idx, void* workspace,
                            hipStream_t stream) {
    if (nq == 0) return hipSuccess;
    const int nseg = workspace ? evdr_topk_segments(nq, n) : 1;
    if (nseg == 1) {
        hipLaunchKernelGGL(topk_kernel, dim3((unsigned)nq), dim3(TK_THREADS), 0, stream, scores, idx_map, n, row_stride,
                           idx_base, k, 1, n, top_scores, top_idx);
        return hipGetLastError();
    }
    const int64_t seg = ((n + nseg - 1) / nseg + 3) & ~(int64_t)3;            // keeps 16-B aligned rows aligned
    float* cs = reinterpret_cast<float*>(workspace);                          // (nq, nseg * k) candidate scores
    int32_t* ci = reinterpret_cast<int32_t*>(cs + nq * nseg * k);             // ... and their reported indices
    hipLaunchKernelGGL(topk_kernel, dim3((unsigned)(nq * nseg)), dim3(TK_THREADS), 0, stream, scores, idx_map, n, row_stride,
                       idx_base, k, nseg, seg, cs, ci);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(topk_kernel, dim3((unsigned)nq), dim3(TK_THREADS), 0, stream, (const float*)cs, (const int32_t*)ci,
                       (int64_t)nseg * k, (int64_t)nseg * k, 0, k, 1, (int64_t)nseg * k, top_scores, top_idx);
    return hipGetLastError();
}
