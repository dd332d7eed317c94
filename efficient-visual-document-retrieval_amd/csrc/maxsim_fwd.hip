// MaxSim forward for gfx950 (MI355X): fused  Q·Pᵀ (MFMA, fp32 accumulate)  ->  masked max over a page's patches  ->
// masked sum over query tokens.   Replaces evaluator/retrieval.py:187-211 of the reference (einsum -> masked_fill ->
// max -> *has -> *qmask -> sum over a materialised 4-D tensor).
//
// Mapping (MI355X-first, not a translation of the four ATen ops; kernels in maxsim_fwd16.hip):
//   * one workgroup = 8 waves (2 per SIMD) = 8*QW queries  x  a chunk of consecutive pages;
//   * every wave keeps its QW queries' tokens RESIDENT in registers as the MFMA B operand (token on the lane), so the
//     max over patches is an in-register v_max3 chain over the accumulator registers + one cross-lane fold per page;
//   * pages stream HBM -> LDS with LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) through a ring with
//     counted vmcnt + raw s_barrier; the LDS image is XOR-swizzled through the per-lane SOURCE address so that the
//     ds_read_b128 A-fragment reads are bank-conflict free;
//   * all 8 waves read the same patch tile from LDS: each page byte is fetched from HBM/L2 once per 8*QW queries, and the
//     block->(query group, page chunk) map puts the query groups that share a page chunk on one XCD so the chunk is
//     served by that XCD's L2;
//   * fp32 inputs are scored to fp32 accuracy as three products of fp16 hi/lo planes (nplanes = 2).
// This file: the choice of queries-per-wave and kernel family for a launch.
#include <atomic>

#include "maxsim_device.h"

// Test / experiment hooks (include/evdr.h "debug hooks"): THREAD-LOCAL, set explicitly through the C ABI -- an override applies
// to the launches the calling thread issues afterwards and to nothing else, so the library holds no shared mutable state that
// concurrent callers could race on.  The library reads no environment variable.
static thread_local int g_fwd_variant = 0;
static thread_local int g_pages_per_block = 0;
static thread_local const char* g_last_fwd_kernel = "";
int evdr_fwd_variant_exchange(int v) { const int old = g_fwd_variant; g_fwd_variant = v; return old; }
int evdr_pages_per_block_exchange(int v) { const int old = g_pages_per_block; g_pages_per_block = v; return old; }
int evdr_pages_per_block_override() { return g_pages_per_block; }
void evdr_note_fwd_kernel(const char* name) { g_last_fwd_kernel = name; }
const char* evdr_last_fwd_kernel_name() { return g_last_fwd_kernel; }

hipError_t evdr_launch_maxsim_fwd(const EvdrFwdParams& pin, int nplanes, bool want_argmax, hipStream_t stream) {
    // 33-40 bf16 queries: 32 at four per wave plus the remainder as its own (HBM-bound, 1-2 per wave) launch, instead of two
    // groups of 17-20 at three per wave with a sixth of the wave slots empty: 9.0 instead of 9.6 ms for 40 queries x 40 k pages
    // (the remainder's pass over the corpus costs less than the idle slots; from 41 queries on the balanced groups win).
    if (nplanes == 1 && !want_argmax && pin.nq > 32 && pin.nq <= 40 && !pin.per_token && pin.qlist == nullptr &&
        g_fwd_variant == 0) {
        EvdrFwdParams a = pin, b = pin;
        a.nq = 32;
        b.nq = pin.nq - 32;
        b.Q = pin.Q + 32 * pin.q_stride;
        if (pin.qmask != nullptr) b.qmask = pin.qmask + (int64_t)32 * pin.lq_total;
        b.out = pin.out + 32 * pin.out_stride;
        if (hipError_t e = evdr_launch_maxsim_fwd(a, nplanes, want_argmax, stream); e != hipSuccess) return e;
        return evdr_launch_maxsim_fwd(b, nplanes, want_argmax, stream);
    }
    EvdrFwdParams p = pin;
    p.ntiles = (p.lp + EVDR_TILE_PATCHES - 1) / EVDR_TILE_PATCHES;
    // more queries per wave = more MFMAs per LDS read, bounded by the 256 VGPRs of a wave at 2 waves per SIMD:
    // 32 per bf16 query, 64 per fp16 hi/lo query, a few more for the running argmax
    const int variant = g_fwd_variant;   // 0 = default dispatch (evdr_debug_set_fwd_variant)
    int qw, waves = 8;
    if (nplanes == 1 && !want_argmax) {
        // Queries per wave from the batch size: a workgroup covers 8 * qw queries, and a workgroup with idle waves takes as
        // long as a full one.  So the batch is cut into ceil(nq / 32) query groups and qw is what ONE group needs per wave:
        // 17-24 queries run three to a wave on all eight waves instead of four to a wave on six (-15..18 % time), 40 queries
        // are two groups of 24 + 16 at three per wave instead of 32 + 8 at four.
        {
            const int groups = (p.nq + 31) / 32;
            const int per_group = (p.nq + groups - 1) / groups;
            qw = (per_group + 7) / 8;
            qw = qw < 1 ? 1 : (qw > 4 ? 4 : qw);
        }
        // 3-12 queries per launch (online retrieval): ONE workgroup of 8 waves leaves the matrix pipes idle around every
        // stage barrier, and with one query per wave a stage is too short to amortise that (63 % MFMA-busy in cycles).  Two
        // independent 4-wave workgroups per CU (80 KiB of LDS each, 4-tile stages) on different page chunks fill each other's
        // gaps: -6..13 % time at 3-4 queries (one per wave), -6..9 % at 5-8 (two per wave), -20 % at 9-12 (three per wave: with two on eight waves
        // a quarter of the waves idled); ONE workgroup still covers all queries, so no page is fetched twice.  (No gain at
        // 13-16 queries with four per wave; at 1024 queries the same split doubles the L2 -> LDS traffic.)
        if (p.nq > 2 && p.nq <= 12 && variant != 30 && p.ntiles >= 4) {
            waves = 4;
            qw = (p.nq + 3) / 4;
        }
#ifdef EVDR_EXPERIMENT
        if (variant == 60 && qw == 4 && p.ntiles >= 8 && !p.per_token) {      // A/B: the same 32 queries per workgroup on four waves of eight
            waves = 4;
            qw = 8;
        }
#endif
    } else if (nplanes == 4) {
        qw = 1;                             // 256-wide embeddings (two column blocks of fp16 hi/lo planes): the query fragments of ONE query fill half the register file
    } else {
        qw = (p.nq > 8) ? 2 : 1;
        // Small launches (a page shard of a multi-GPU training step: 32 queries x 63 pages = 126 workgroups of 16 queries on 256
        // CUs): one query per wave makes twice the workgroups of half the matrix work each -- student forward + arg-max 20.9 ->
        // 14.7 us, teacher forward 59.4 -> 40.8 us at 63 pages; from ~100 pages on (>= 200 workgroups) the two forms are level
        // and two queries per wave move half the LDS bytes per FLOP.  Same bits.  Variant 36 keeps two per wave (A/B, tests).
        if (qw == 2 && variant != 36 && (int64_t)((p.nq + 15) / 16) * p.np <= 128) qw = 1;
#ifdef EVDR_EXPERIMENT
        if (variant == 13) qw = 2;          // dispatched as one query per wave on 4-wave workgroups (see maxsim_fwd16.hip)
        if (variant == 35) qw = 1;          // A/B: one query per wave (8 per workgroup, twice the query groups, half the prologue bytes per workgroup)
#endif
    }
    // HBM-bound launches (a handful of queries) want the refill in flight as early as possible: +3 % at 1-4 queries;
    // everything else hides the refill's address work under MFMAs: +2..4 %
    p.inblock_refill = p.nq > 4 ? 1 : 0;
    return evdr_launch_maxsim_fwd16(p, qw, waves, nplanes, want_argmax, variant, stream);
}
