// C-ABI entry points (include/evdr.h).  Argument checking + workspace carving + kernel dispatch.
// Nothing here allocates, frees or synchronises (graph-capture safe); errors come back as status codes.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>

#include "evdr_common.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    return fail(EVDR_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

inline int64_t ntiles_of(int64_t lp) { return (lp + EVDR_TILE_PATCHES - 1) / EVDR_TILE_PATCHES; }

struct FwdWorkspace {
    size_t tilemask_off, pageflags_off, qlist_off, amax_off, qplanes_off, pplanes_off, total;
};

FwdWorkspace carve_fwd(int64_t nq, int64_t lq, int64_t np, int64_t lp, int dtype) {
    FwdWorkspace w{};
    size_t off = 0;
    w.tilemask_off = off;
    off = align_up(off + (size_t)np * ntiles_of(lp) * 4);
    w.pageflags_off = off;
    off = align_up(off + (size_t)np * 4);
    w.qlist_off = off;                                  // later token slices: [0] = count, [1..nq] = compacted query indices
    if (lq > 32) off = align_up(off + (size_t)(nq + 1) * 4);
    w.amax_off = off;                                   // [0] = Q's absmax bits, [1] = P's
    if (dtype == EVDR_F32) off = align_up(off + 2 * sizeof(uint32_t));
    w.qplanes_off = off;
    if (dtype == EVDR_F32) off = align_up(off + (size_t)2 * nq * lq * EVDR_D * 2);
    w.pplanes_off = off;
    if (dtype == EVDR_F32) off = align_up(off + (size_t)2 * np * lp * EVDR_D * 2);
    w.total = off;
    return w;
}

int check_common(int64_t nq, int64_t lq, int64_t np, int64_t lp) {
    if (nq < 0 || lq < 0 || np < 0 || lp < 0) return fail(EVDR_ERR_ARG, "negative size");
    if (nq > INT32_MAX || np > INT32_MAX || lq > 65535 || lp > 65535)
        return fail(EVDR_ERR_SHAPE, "size out of range (nq=%lld lq=%lld np=%lld lp=%lld)", (long long)nq,
                    (long long)lq, (long long)np, (long long)lp);
    if (nq * lq > INT32_MAX) return fail(EVDR_ERR_SHAPE, "nq * lq = %lld exceeds 2^31 - 1 (kernels index (query, token) pairs in 32 bits)",
                                         (long long)(nq * lq));
    return EVDR_OK;
}

// Runs the forward over 32-token slices of the queries (one slice for Lq <= 32).
// Single-token queries (Lq = 1: the "virtual queries" of mainv3_iter_liscore_QA_hardtoken.py:428-434) would leave 31 of
// the 32 token lanes of the MFMA tile empty; dense (nq, 1, 128) rows ARE the layout of ceil(nq/32) queries of 32 tokens,
// so they are scored as such with one output row per token (EvdrFwdParams::per_token): 32x less MFMA work.
int run_fwd(const uint16_t* Qp, int64_t q_stride, int64_t q_plane_stride, const uint16_t* Pp, int64_t p_stride,
            int64_t p_plane_stride, const uint8_t* qmask, const uint32_t* tilemask, const uint32_t* pageflags,
            float* out, int64_t out_stride, uint16_t* argmax, int64_t nq, int64_t lq, int64_t np, int64_t lp,
            int nplanes, const uint32_t* q_amax, const uint32_t* p_amax, int32_t* qlist_ws, hipStream_t stream,
            const int32_t* qsel = nullptr, const int32_t* qsel_count = nullptr) {
    const bool pack = (lq == 1 && nq > 1 && q_stride == EVDR_D);
    for (int64_t tok0 = 0; tok0 < lq; tok0 += 32) {
        EvdrFwdParams p{};
        p.Q = Qp;
        p.q_stride = pack ? 32 * EVDR_D : q_stride;
        p.q_plane_stride = q_plane_stride;
        p.P = Pp;
        p.p_stride = p_stride;
        p.p_plane_stride = p_plane_stride;
        p.q_amax = q_amax;
        p.p_amax = p_amax;
        p.qmask = qmask;
        p.tilemask = tilemask;
        p.pageflags = pageflags;
        p.out = out;
        p.out_stride = out_stride;
        p.argmax = argmax;
        p.nq = (int)(pack ? (nq + 31) / 32 : nq);
        p.lq = pack ? 32 : (int)((lq - tok0 < 32) ? (lq - tok0) : 32);
        p.np = (int)np;
        p.lp = (int)lp;
        p.tok0 = (int)tok0;
        p.lq_total = pack ? 32 : (int)lq;
        p.per_token = pack ? nq : 0;
        p.accumulate = tok0 > 0 ? 1 : 0;
        if (tok0 > 0 && qlist_ws && qmask && !argmax) {
            // queries padded to the longest of a set: only those with a valid token in this slice are scored again
            hipError_t eq = evdr_launch_build_qlist(qmask, nq, lq, tok0, qlist_ws + 1, qlist_ws, stream);
            if (eq != hipSuccess) return hip_fail(eq, "build_qlist launch");
            p.qlist = qlist_ws + 1;
            p.qcount = qlist_ws;
        }
        if (qsel != nullptr) {                          // evdr_maxsim_fwd_prepared_subset: one slice, the caller's device-side list
            p.qlist = qsel;
            p.qcount = qsel_count;
        }
        hipError_t e = evdr_launch_maxsim_fwd(p, nplanes, argmax != nullptr, stream);
        if (e != hipSuccess) return hip_fail(e, "maxsim_fwd launch");
    }
    return EVDR_OK;
}

}  // namespace

extern "C" {

int evdr_version(void) { return EVDR_VERSION_NUM; }

const char* evdr_last_error(void) { return g_err; }

int evdr_debug_set_fwd_variant(int variant) { return evdr_fwd_variant_exchange(variant); }

int evdr_debug_set_pages_per_block(int pages) { return evdr_pages_per_block_exchange(pages < 0 ? 0 : pages); }

const char* evdr_last_fwd_kernel(void) { return evdr_last_fwd_kernel_name(); }

int evdr_pack_pmask(const uint8_t* pmask, int64_t np, int64_t lp, uint32_t* tilemask, uint32_t* pageflags,
                    void* hip_stream) {
    if (int rc = check_common(0, 0, np, lp)) return rc;
    if (!tilemask || !pageflags) return fail(EVDR_ERR_ARG, "evdr_pack_pmask: null output");
    if (np == 0) return EVDR_OK;
    if (lp == 0) return fail(EVDR_ERR_SHAPE, "evdr_pack_pmask: lp == 0");
    hipError_t e = evdr_launch_pack_pmask(pmask, np, lp, tilemask, pageflags, (hipStream_t)hip_stream);
    return e == hipSuccess ? EVDR_OK : hip_fail(e, "pack_pmask launch");
}

int evdr_split_f32(const float* x, int64_t rows, uint16_t* planes, uint32_t* amax_bits, void* hip_stream) {
    if (rows < 0) return fail(EVDR_ERR_ARG, "evdr_split_f32: negative rows");
    if (!amax_bits) return fail(EVDR_ERR_ARG, "evdr_split_f32: null amax_bits");
    if (rows > 0 && (!x || !planes)) return fail(EVDR_ERR_ARG, "evdr_split_f32: null pointer");
    hipError_t e = evdr_launch_split_f32(x, rows, planes, amax_bits, (hipStream_t)hip_stream);
    return e == hipSuccess ? EVDR_OK : hip_fail(e, "split_f32 launch");
}

int evdr_split_f32_segments(const float* x, int64_t rows, int64_t seg_rows, uint16_t* planes, uint32_t* amax_bits, void* hip_stream) {
    if (rows < 0) return fail(EVDR_ERR_ARG, "evdr_split_f32_segments: negative rows");
    if (seg_rows < 1 || seg_rows > 2048) return fail(EVDR_ERR_ARG, "evdr_split_f32_segments: seg_rows outside 1..2048");
    if ((rows + seg_rows - 1) / seg_rows > 65535) return fail(EVDR_ERR_ARG, "evdr_split_f32_segments: more than 65535 segments");
    if (rows > 0 && (!x || !planes || !amax_bits)) return fail(EVDR_ERR_ARG, "evdr_split_f32_segments: null pointer");
    hipError_t e = evdr_launch_split_f32_segments(x, rows, seg_rows, planes, amax_bits, (hipStream_t)hip_stream);
    return e == hipSuccess ? EVDR_OK : hip_fail(e, "split_f32_segments launch");
}

int evdr_flag_nonfinite(const void* P, int dtype, const uint8_t* pmask, int64_t np, int64_t lp, int64_t p_stride,
                        uint32_t* pageflags, void* hip_stream) {
    if (int rc = check_common(0, 0, np, lp)) return rc;
    if (dtype != EVDR_F32 && dtype != EVDR_BF16 && dtype != EVDR_F16) return fail(EVDR_ERR_ARG, "dtype must be EVDR_F32, EVDR_BF16 or EVDR_F16");
    if (np == 0 || lp == 0) return EVDR_OK;
    if (!P || !pageflags) return fail(EVDR_ERR_ARG, "evdr_flag_nonfinite: null pointer");
    if (p_stride < lp * EVDR_D) return fail(EVDR_ERR_ARG, "p_stride smaller than a page");
    hipError_t e = evdr_launch_flag_nonfinite(P, dtype, pmask, np, lp, p_stride, pageflags, (hipStream_t)hip_stream);
    return e == hipSuccess ? EVDR_OK : hip_fail(e, "flag_nonfinite launch");
}

size_t evdr_maxsim_fwd_workspace(int64_t nq, int64_t lq, int64_t np, int64_t lp, int dtype) {
    if (nq < 0 || lq < 0 || np < 0 || lp < 0) return 0;
    return carve_fwd(nq, lq, np, lp, dtype).total;
}

int evdr_maxsim_fwd(const void* Q, const void* P, const uint8_t* qmask, const uint8_t* pmask, float* out,
                    uint16_t* argmax_or_null, int64_t nq, int64_t lq, int64_t np, int64_t lp, int64_t d, int dtype,
                    const int64_t* strides_or_null, void* workspace, size_t workspace_bytes, void* hip_stream) {
    if (int rc = check_common(nq, lq, np, lp)) return rc;
    if (d != EVDR_D)
        return fail(EVDR_ERR_SHAPE, "embedding width %lld: this entry takes width 128 (widths up to 256: evdr_maxsim_fwd_prepared with nplanes = 4)", (long long)d);
    if (dtype != EVDR_F32 && dtype != EVDR_BF16) return fail(EVDR_ERR_ARG, "dtype must be EVDR_F32 or EVDR_BF16");
    if (nq == 0 || np == 0) return EVDR_OK;                      // empty score matrix
    if (lq == 0 || lp == 0) return fail(EVDR_ERR_SHAPE, "zero-length token axis (lq=%lld lp=%lld)", (long long)lq, (long long)lp);
    if (!Q || !P || !out) return fail(EVDR_ERR_ARG, "evdr_maxsim_fwd: null Q/P/out");
    const int64_t q_stride = strides_or_null ? strides_or_null[0] : lq * EVDR_D;
    const int64_t p_stride = strides_or_null ? strides_or_null[1] : lp * EVDR_D;
    if (q_stride < lq * EVDR_D || p_stride < lp * EVDR_D) return fail(EVDR_ERR_ARG, "strides smaller than a row block");
    const FwdWorkspace w = carve_fwd(nq, lq, np, lp, dtype);
    if (!workspace || workspace_bytes < w.total)
        return fail(EVDR_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", w.total, workspace_bytes);
    hipStream_t stream = (hipStream_t)hip_stream;
    char* ws = (char*)workspace;
    uint32_t* tilemask = (uint32_t*)(ws + w.tilemask_off);
    uint32_t* pageflags = (uint32_t*)(ws + w.pageflags_off);
    hipError_t e = evdr_launch_pack_pmask(pmask, np, lp, tilemask, pageflags, stream);
    if (e != hipSuccess) return hip_fail(e, "pack_pmask launch");
    if (dtype == EVDR_BF16) {
        // NaN / Inf in a valid patch -> page flag bit 3 (one extra read of P; a resident corpus pays it once, evdr_flag_nonfinite)
        if ((e = evdr_launch_flag_nonfinite(P, 1, pmask, np, lp, p_stride, pageflags, stream)) != hipSuccess)
            return hip_fail(e, "flag_nonfinite launch");
        return run_fwd((const uint16_t*)Q, q_stride, 0, (const uint16_t*)P, p_stride, 0, qmask, tilemask, pageflags, out,
                       np, argmax_or_null, nq, lq, np, lp, 1, nullptr, nullptr, lq > 32 ? (int32_t*)(ws + w.qlist_off) : nullptr, stream);
    }
    if (q_stride != lq * EVDR_D || p_stride != lp * EVDR_D)
        return fail(EVDR_ERR_ARG, "fp32 inputs must be dense (make them contiguous before the call)");
    uint16_t* qpl = (uint16_t*)(ws + w.qplanes_off);
    uint16_t* ppl = (uint16_t*)(ws + w.pplanes_off);
    uint32_t* amax = (uint32_t*)(ws + w.amax_off);
    if ((e = evdr_launch_split_f32((const float*)Q, nq * lq, qpl, amax, stream)) != hipSuccess) return hip_fail(e, "split Q");
    if ((e = evdr_launch_split_f32_pages((const float*)P, np * lp, ppl, amax + 1, pmask, lp, pageflags, stream)) != hipSuccess)
        return hip_fail(e, "split P");
    return run_fwd(qpl, lq * EVDR_D, nq * lq * EVDR_D, ppl, lp * EVDR_D, np * lp * EVDR_D, qmask, tilemask, pageflags, out,
                   np, argmax_or_null, nq, lq, np, lp, 2, amax, amax + 1, lq > 32 ? (int32_t*)(ws + w.qlist_off) : nullptr, stream);
}

int evdr_maxsim_fwd_prepared(const uint16_t* Qplanes, const uint16_t* Pplanes, const uint8_t* qmask,
                             const uint32_t* tilemask, const uint32_t* pageflags, float* out, int64_t out_stride,
                             uint16_t* argmax_or_null, int64_t nq, int64_t lq, int64_t np, int64_t lp, int nplanes,
                             int64_t p_stride, int64_t p_plane_stride, const uint32_t* q_amax_or_null,
                             const uint32_t* p_amax_or_null, int32_t* qlist_ws_or_null, void* hip_stream) {
    if (int rc = check_common(nq, lq, np, lp)) return rc;
    if (nplanes != 1 && nplanes != 2 && nplanes != 4)
        return fail(EVDR_ERR_ARG, "nplanes must be 1 (bf16), 2 (fp16 hi/lo) or 4 (fp16 hi/lo x two 128-column blocks: width 256)");
    if (nq == 0 || np == 0) return EVDR_OK;
    if (lq == 0 || lp == 0) return fail(EVDR_ERR_SHAPE, "zero-length token axis");
    if (!Qplanes || !Pplanes || !tilemask || !pageflags || !out) return fail(EVDR_ERR_ARG, "evdr_maxsim_fwd_prepared: null pointer");
    if (out_stride < np || p_stride < lp * EVDR_D) return fail(EVDR_ERR_ARG, "stride smaller than the row it spans");
    return run_fwd(Qplanes, lq * EVDR_D, nq * lq * EVDR_D, Pplanes, p_stride, p_plane_stride, qmask, tilemask, pageflags, out,
                   out_stride, argmax_or_null, nq, lq, np, lp, nplanes, nplanes >= 2 ? q_amax_or_null : nullptr,
                   nplanes >= 2 ? p_amax_or_null : nullptr, lq > 32 ? qlist_ws_or_null : nullptr, (hipStream_t)hip_stream);
}

int evdr_maxsim_fwd_prepared_subset(const uint16_t* Qplanes, const uint16_t* Pplanes, const uint8_t* qmask,
                                    const uint32_t* tilemask, const uint32_t* pageflags, float* out, int64_t out_stride,
                                    int64_t nq, int64_t lq, int64_t np, int64_t lp, int nplanes, int64_t p_stride,
                                    int64_t p_plane_stride, const uint32_t* q_amax_or_null, const uint32_t* p_amax_or_null,
                                    const int32_t* qsel, const int32_t* qsel_count, void* hip_stream) {
    if (int rc = check_common(nq, lq, np, lp)) return rc;
    if (nplanes != 1 && nplanes != 2) return fail(EVDR_ERR_ARG, "nplanes must be 1 (bf16) or 2 (fp16 hi/lo)");
    if (nq == 0 || np == 0) return EVDR_OK;
    if (lq == 0 || lp == 0) return fail(EVDR_ERR_SHAPE, "zero-length token axis");
    if (lq > 32 || (lq == 1 && nq > 1)) return fail(EVDR_ERR_SHAPE, "evdr_maxsim_fwd_prepared_subset: lq must be 2..32 (or one single-token query)");
    if (!Qplanes || !Pplanes || !tilemask || !pageflags || !out || !qsel || !qsel_count)
        return fail(EVDR_ERR_ARG, "evdr_maxsim_fwd_prepared_subset: null pointer");
    if (out_stride < np || p_stride < lp * EVDR_D) return fail(EVDR_ERR_ARG, "stride smaller than the row it spans");
    return run_fwd(Qplanes, lq * EVDR_D, nq * lq * EVDR_D, Pplanes, p_stride, p_plane_stride, qmask, tilemask, pageflags, out,
                   out_stride, nullptr, nq, lq, np, lp, nplanes, nplanes == 2 ? q_amax_or_null : nullptr,
                   nplanes == 2 ? p_amax_or_null : nullptr, nullptr, (hipStream_t)hip_stream, qsel, qsel_count);
}

static int check_qcache(const EvdrQCache* c, const char* who) {
    if (!c) return fail(EVDR_ERR_ARG, "%s: null cache", who);
    if (!c->slots || !c->ent_hash || !c->ent_k || !c->ent_q || !c->ent_mask || !c->ent_scores || !c->n_entries)
        return fail(EVDR_ERR_ARG, "%s: null pointer in the cache struct", who);
    if (c->capacity < 1 || c->capacity > INT32_MAX - 2 || c->lq < 1 || c->np < 1 || c->row_bytes < 4 || (c->row_bytes & 3))
        return fail(EVDR_ERR_ARG, "%s: bad cache geometry", who);
    if (c->n_slots < 2 * c->capacity || (c->n_slots & (c->n_slots - 1)) || c->n_slots > INT32_MAX)
        return fail(EVDR_ERR_ARG, "%s: n_slots must be a power of two >= 2 * capacity", who);
    return EVDR_OK;
}

size_t evdr_qcache_workspace(int64_t nq) {
    if (nq < 0) return 0;
    return align_up((size_t)nq * 8) + 2 * align_up((size_t)nq * 4) + 256;      // hashes | hit | qsel | {miss count, ticket}
}

int evdr_maxsim_fwd_prepared_cached(const EvdrQCache* cache, const void* Qrows, const uint16_t* Qplanes, const uint16_t* Pplanes,
                                    const uint8_t* qmask, const uint32_t* tilemask, const uint32_t* pageflags, float* out,
                                    int64_t out_stride, int64_t nq, int64_t lp, int nplanes, int64_t p_stride, int64_t p_plane_stride,
                                    const uint32_t* q_amax_or_null, const uint32_t* p_amax_or_null, void* workspace,
                                    size_t workspace_bytes, void* hip_stream) {
    if (int rc = check_qcache(cache, "evdr_maxsim_fwd_prepared_cached")) return rc;
    const int64_t lq = cache->lq, np = cache->np;
    if (int rc = check_common(nq, lq, np, lp)) return rc;
    if (nplanes != 1 && nplanes != 2) return fail(EVDR_ERR_ARG, "nplanes must be 1 (bf16) or 2 (fp16 hi/lo)");
    if (nq == 0) return EVDR_OK;
    if (lp == 0) return fail(EVDR_ERR_SHAPE, "zero-length token axis");
    if (lq > 32 || lq < 2) return fail(EVDR_ERR_SHAPE, "evdr_maxsim_fwd_prepared_cached: the cache's lq must be 2..32");
    if (!Qrows || !Qplanes || !Pplanes || !tilemask || !pageflags || !out) return fail(EVDR_ERR_ARG, "evdr_maxsim_fwd_prepared_cached: null pointer");
    if (out_stride < np || p_stride < lp * EVDR_D) return fail(EVDR_ERR_ARG, "stride smaller than the row it spans");
    if (!workspace || workspace_bytes < evdr_qcache_workspace(nq))
        return fail(EVDR_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", evdr_qcache_workspace(nq), workspace_bytes);
    hipStream_t stream = (hipStream_t)hip_stream;
    char* ws = (char*)workspace;
    uint64_t* hashes = (uint64_t*)ws;
    int32_t* hit = (int32_t*)(ws + align_up((size_t)nq * 8));
    int32_t* qsel = (int32_t*)(ws + align_up((size_t)nq * 8) + align_up((size_t)nq * 4));
    int32_t* count = (int32_t*)(ws + align_up((size_t)nq * 8) + 2 * align_up((size_t)nq * 4));
    uint32_t* ticket = (uint32_t*)(count + 1);
    const uint32_t* qa = nplanes == 2 ? q_amax_or_null : nullptr;
    hipError_t e = evdr_launch_qcache_lookup_plan(*cache, Qrows, qmask, qa, nq, hashes, hit, qsel, count, ticket, stream);
    if (e != hipSuccess) return hip_fail(e, "qcache_lookup_plan launch");
    if (int rc = run_fwd(Qplanes, lq * EVDR_D, nq * lq * EVDR_D, Pplanes, p_stride, p_plane_stride, qmask, tilemask, pageflags, out,
                         out_stride, nullptr, nq, lq, np, lp, nplanes, qa, nplanes == 2 ? p_amax_or_null : nullptr, nullptr, stream,
                         qsel, count))
        return rc;
    e = evdr_launch_qcache_exchange(*cache, Qrows, qmask, nq, hit, out, out_stride, stream);
    return e == hipSuccess ? EVDR_OK : hip_fail(e, "qcache_exchange launch");
}

int evdr_maxsim_bwd(const float* g, const float* Q, const uint8_t* qmask, const uint8_t* pmask, const uint16_t* argmax,
                    float* dP, int64_t nq, int64_t lq, int64_t np, int64_t lp, int64_t d, void* hip_stream) {
    if (int rc = check_common(nq, lq, np, lp)) return rc;
    if (d != EVDR_D) return fail(EVDR_ERR_SHAPE, "embedding width %lld unsupported", (long long)d);
    if (np == 0 || lp == 0) return EVDR_OK;
    if (!dP) return fail(EVDR_ERR_ARG, "evdr_maxsim_bwd: null dP");
    if (nq > 0 && lq > 0 && (!g || !Q || !argmax)) return fail(EVDR_ERR_ARG, "evdr_maxsim_bwd: null g/Q/argmax");
    hipError_t e = evdr_launch_maxsim_bwd(g, Q, qmask, pmask, argmax, dP, nq, lq, np, lp, (hipStream_t)hip_stream);
    return e == hipSuccess ? EVDR_OK : hip_fail(e, "maxsim_bwd launch");
}

size_t evdr_maxsim_bwd_q_workspace(int64_t nq, int64_t lq, int64_t np, int64_t lp) {
    if (nq < 0 || lq < 0 || np < 0 || lp < 0) return 0;
    return align_up((size_t)np * ntiles_of(lp) * 4) + align_up((size_t)np * 4) +
           align_up((size_t)evdr_bwd_q_segments(nq * lq, np) * nq * lq * EVDR_D * sizeof(float));
}

int evdr_maxsim_bwd_q(const float* g, const float* P, const uint8_t* qmask, const uint8_t* pmask, const uint16_t* argmax,
                      float* dQ, int64_t nq, int64_t lq, int64_t np, int64_t lp, int64_t d, void* workspace,
                      size_t workspace_bytes, void* hip_stream) {
    if (int rc = check_common(nq, lq, np, lp)) return rc;
    if (d != EVDR_D) return fail(EVDR_ERR_SHAPE, "embedding width %lld unsupported", (long long)d);
    if (nq == 0 || lq == 0) return EVDR_OK;
    if (!dQ) return fail(EVDR_ERR_ARG, "evdr_maxsim_bwd_q: null dQ");
    if (np > 0 && (!g || !P || !argmax)) return fail(EVDR_ERR_ARG, "evdr_maxsim_bwd_q: null g/P/argmax");
    if (np > 0 && lp == 0) return fail(EVDR_ERR_SHAPE, "lp == 0");
    if (!workspace || workspace_bytes < evdr_maxsim_bwd_q_workspace(nq, lq, np, lp))
        return fail(EVDR_ERR_WORKSPACE, "workspace too small: need %zu bytes", evdr_maxsim_bwd_q_workspace(nq, lq, np, lp));
    hipStream_t stream = (hipStream_t)hip_stream;
    uint32_t* tilemask = (uint32_t*)workspace;
    uint32_t* pageflags = (uint32_t*)((char*)workspace + align_up((size_t)np * ntiles_of(lp) * 4));
    hipError_t e = hipSuccess;
    if (np > 0 && (e = evdr_launch_pack_pmask(pmask, np, lp, tilemask, pageflags, stream)) != hipSuccess)
        return hip_fail(e, "pack_pmask launch");
    float* partials = (float*)((char*)workspace + align_up((size_t)np * ntiles_of(lp) * 4) + align_up((size_t)np * 4));
    e = evdr_launch_maxsim_bwd_q(g, P, qmask, pageflags, argmax, dQ, partials, nq, lq, np, lp, stream);
    return e == hipSuccess ? EVDR_OK : hip_fail(e, "maxsim_bwd_q launch");
}

int evdr_maxsim_bwd_adamw(const float* g, const float* Q, const uint8_t* qmask, const uint8_t* pmask, const uint16_t* argmax,
                          float* x, float* exp_avg, float* exp_avg_sq, int64_t nq, int64_t lq, int64_t np, int64_t lp,
                          int64_t d, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                          float l2_eps, const void* adamw_state_or_null, void* hip_stream) {
    return evdr_maxsim_bwd_adamw_planes(g, Q, qmask, pmask, argmax, x, exp_avg, exp_avg_sq, nq, lq, np, lp, d, lr, beta1, beta2,
                                        eps, weight_decay, step, l2_eps, adamw_state_or_null, nullptr, nullptr, nullptr,
                                        hip_stream);
}

int evdr_maxsim_bwd_adamw_planes(const float* g, const float* Q, const uint8_t* qmask, const uint8_t* pmask,
                                 const uint16_t* argmax, float* x, float* exp_avg, float* exp_avg_sq, int64_t nq, int64_t lq,
                                 int64_t np, int64_t lp, int64_t d, double lr, double beta1, double beta2, double eps,
                                 double weight_decay, int64_t step, float l2_eps, const void* adamw_state_or_null,
                                 void* next_planes_or_null, uint32_t* next_amax_or_null, uint32_t* pageflags_or_null,
                                 void* hip_stream) {
    if (int rc = check_common(nq, lq, np, lp)) return rc;
    if (d != EVDR_D) return fail(EVDR_ERR_SHAPE, "embedding width %lld unsupported", (long long)d);
    if (!adamw_state_or_null && step < 1) return fail(EVDR_ERR_ARG, "step must be >= 1 (1 for the first update)");
    if (np == 0 || lp == 0) return EVDR_OK;
    if (!x || !exp_avg || !exp_avg_sq) return fail(EVDR_ERR_ARG, "evdr_maxsim_bwd_adamw: null parameter/state");
    if (nq > 0 && lq > 0 && (!g || !Q || !argmax)) return fail(EVDR_ERR_ARG, "evdr_maxsim_bwd_adamw: null g/Q/argmax");
    if (next_planes_or_null && !next_amax_or_null) return fail(EVDR_ERR_ARG, "evdr_maxsim_bwd_adamw_planes: planes without their absmax word");
    const double bc1 = 1.0 - pow(beta1, (double)(step < 1 ? 1 : step));
    const double bc2 = 1.0 - pow(beta2, (double)(step < 1 ? 1 : step));
    hipError_t e = evdr_launch_maxsim_bwd_adamw(g, Q, qmask, pmask, argmax, x, exp_avg, exp_avg_sq, nq, lq, np, lp, lr, beta1,
                                                beta2, eps, weight_decay, bc1, sqrt(bc2), l2_eps,
                                                adamw_state_or_null, next_planes_or_null, next_amax_or_null,
                                                pageflags_or_null, (hipStream_t)hip_stream);
    return e == hipSuccess ? EVDR_OK : hip_fail(e, "maxsim_bwd_adamw launch");
}

int evdr_adamw_step(const float* grad, float* x, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                    double beta2, double eps, double weight_decay, int64_t step, void* hip_stream) {
    if (n < 0) return fail(EVDR_ERR_ARG, "negative size");
    if (n == 0) return EVDR_OK;
    if (!grad || !x || !exp_avg || !exp_avg_sq) return fail(EVDR_ERR_ARG, "evdr_adamw_step: null pointer");
    if (((uintptr_t)grad | (uintptr_t)x | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15)
        return fail(EVDR_ERR_ARG, "evdr_adamw_step: tensors must be 16-byte aligned");
    const double bc1 = 1.0 - pow(beta1, (double)(step < 1 ? 1 : step));
    const double bc2 = 1.0 - pow(beta2, (double)(step < 1 ? 1 : step));
    hipError_t e = evdr_launch_adamw(grad, x, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, sqrt(bc2),
                                     (hipStream_t)hip_stream);
    return e == hipSuccess ? EVDR_OK : hip_fail(e, "adamw launch");
}

int evdr_adamw_advance(void* adamw_state, double beta1, double beta2, void* hip_stream) {
    if (!adamw_state) return fail(EVDR_ERR_ARG, "evdr_adamw_advance: null state");
    hipError_t e = evdr_launch_adamw_advance(adamw_state, beta1, beta2, (hipStream_t)hip_stream);
    return e == hipSuccess ? EVDR_OK : hip_fail(e, "adamw_advance launch");
}

int evdr_l2norm_fwd(const float* x, const uint8_t* rowmask_or_null, int64_t rows, int64_t d, float eps, float* y,
                    float* norm_or_null, void* hip_stream) {
    if (rows < 0) return fail(EVDR_ERR_ARG, "negative rows");
    if (d != EVDR_D) return fail(EVDR_ERR_SHAPE, "embedding width %lld unsupported", (long long)d);
    if (rows == 0) return EVDR_OK;
    if (!x || !y) return fail(EVDR_ERR_ARG, "evdr_l2norm_fwd: null pointer");
    hipError_t e = evdr_launch_l2norm_fwd(x, rowmask_or_null, rows, eps, y, norm_or_null, nullptr, nullptr, nullptr, 1,
                                          (hipStream_t)hip_stream);
    return e == hipSuccess ? EVDR_OK : hip_fail(e, "l2norm_fwd launch");
}

int evdr_l2norm_fwd_split(const float* x, const uint8_t* rowmask_or_null, int64_t rows, int64_t d, float eps, float* y_or_null,
                          float* norm_or_null, uint16_t* planes, uint32_t* amax_bits, uint32_t* pageflags_or_null,
                          int64_t rows_per_page, void* hip_stream) {
    if (pageflags_or_null && rows_per_page < 1) return fail(EVDR_ERR_ARG, "evdr_l2norm_fwd_split: rows_per_page must be >= 1");
    if (rows < 0) return fail(EVDR_ERR_ARG, "negative rows");
    if (d != EVDR_D) return fail(EVDR_ERR_SHAPE, "embedding width %lld unsupported", (long long)d);
    if (!amax_bits) return fail(EVDR_ERR_ARG, "evdr_l2norm_fwd_split: null amax_bits");
    if (rows == 0) return EVDR_OK;
    if (!x || !planes) return fail(EVDR_ERR_ARG, "evdr_l2norm_fwd_split: null pointer");
    hipError_t e = evdr_launch_l2norm_fwd(x, rowmask_or_null, rows, eps, y_or_null, norm_or_null, planes, amax_bits,
                                          pageflags_or_null, rows_per_page, (hipStream_t)hip_stream);
    return e == hipSuccess ? EVDR_OK : hip_fail(e, "l2norm_fwd_split launch");
}

int evdr_l2norm_bwd(const float* gy, const float* x, const uint8_t* rowmask_or_null, const float* norm, int64_t rows,
                    int64_t d, float eps, float* dx, void* hip_stream) {
    if (rows < 0) return fail(EVDR_ERR_ARG, "negative rows");
    if (d != EVDR_D) return fail(EVDR_ERR_SHAPE, "embedding width %lld unsupported", (long long)d);
    if (rows == 0) return EVDR_OK;
    if (!gy || !x || !norm || !dx) return fail(EVDR_ERR_ARG, "evdr_l2norm_bwd: null pointer");
    hipError_t e = evdr_launch_l2norm_bwd(gy, x, rowmask_or_null, norm, rows, eps, dx, (hipStream_t)hip_stream);
    return e == hipSuccess ? EVDR_OK : hip_fail(e, "l2norm_bwd launch");
}

size_t evdr_topk_workspace(int64_t nq, int64_t n, int k) {
    if (nq <= 0 || n <= 0 || k < 1) return 0;
    const int nseg = evdr_topk_segments(nq, n);
    return nseg == 1 ? 0 : align_up((size_t)nq * nseg * k * 8);
}

int evdr_topk(const float* scores, const int32_t* idx_map_or_null, int64_t nq, int64_t n, int64_t row_stride,
              int32_t idx_base, int k, float* top_scores, int32_t* top_idx, void* workspace_or_null, size_t workspace_bytes,
              void* hip_stream) {
    if (nq < 0 || n < 0) return fail(EVDR_ERR_ARG, "negative size");
    if (k < 1 || k > EVDR_TOPK_MAX) return fail(EVDR_ERR_ARG, "k=%d outside 1..%d", k, EVDR_TOPK_MAX);
    if (nq == 0) return EVDR_OK;
    if (!top_scores || !top_idx || (n > 0 && !scores)) return fail(EVDR_ERR_ARG, "evdr_topk: null pointer");
    if (row_stride < n) return fail(EVDR_ERR_ARG, "row_stride < n");
    if (workspace_or_null && workspace_bytes < evdr_topk_workspace(nq, n, k)) workspace_or_null = nullptr;   // too small: one level
    hipError_t e = evdr_launch_topk(scores, idx_map_or_null, nq, n, row_stride, idx_base, k, top_scores, top_idx,
                                    workspace_or_null, (hipStream_t)hip_stream);
    return e == hipSuccess ? EVDR_OK : hip_fail(e, "topk launch");
}

size_t evdr_maxsim_topk_workspace(int64_t nq, int64_t np) {
    if (nq < 0 || np < 0) return 0;
    return align_up((size_t)nq * np * sizeof(float)) + evdr_topk_workspace(nq, np, EVDR_TOPK_MAX);
}

int evdr_maxsim_topk(const uint16_t* Qplanes, const uint16_t* Pplanes, const uint8_t* qmask, const uint32_t* tilemask,
                     const uint32_t* pageflags, int64_t nq, int64_t lq, int64_t np, int64_t lp, int nplanes,
                     int64_t p_stride, int64_t p_plane_stride, const uint32_t* q_amax_or_null,
                     const uint32_t* p_amax_or_null, int32_t idx_base, int k, float* top_scores,
                     int32_t* top_idx, void* workspace, size_t workspace_bytes, void* hip_stream) {
    if (!workspace || workspace_bytes < evdr_maxsim_topk_workspace(nq, np))
        return fail(EVDR_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu",
                    evdr_maxsim_topk_workspace(nq, np), workspace_bytes);
    float* scores = (float*)workspace;
    int rc = evdr_maxsim_fwd_prepared(Qplanes, Pplanes, qmask, tilemask, pageflags, scores, np, nullptr, nq, lq, np, lp,
                                      nplanes, p_stride, p_plane_stride, q_amax_or_null, p_amax_or_null, nullptr, hip_stream);
    if (rc != EVDR_OK) return rc;
    char* tkws = (char*)workspace + align_up((size_t)nq * np * sizeof(float));
    return evdr_topk(scores, nullptr, nq, np, np, idx_base, k, top_scores, top_idx, tkws, evdr_topk_workspace(nq, np, EVDR_TOPK_MAX),
                     hip_stream);
}

int evdr_infonce_distill_fwd_bwd(const float* score_s, const float* score_t, int64_t b, int64_t n, float temperature,
                                 float* loss, float* dscore_or_null, float* row_loss, void* hip_stream) {
    if (b < 0 || n < 0) return fail(EVDR_ERR_ARG, "negative size");
    if (!(temperature > 0.f)) return fail(EVDR_ERR_ARG, "temperature must be > 0");
    if (b == 0 || n == 0) return fail(EVDR_ERR_SHAPE, "empty score matrix (b=%lld n=%lld)", (long long)b, (long long)n);
    if (!score_s || !score_t || !loss || !row_loss) return fail(EVDR_ERR_ARG, "evdr_infonce_distill_fwd_bwd: null pointer");
    hipError_t e = evdr_launch_infonce(score_s, score_t, b, n, temperature, loss, dscore_or_null, row_loss,
                                       (hipStream_t)hip_stream);
    return e == hipSuccess ? EVDR_OK : hip_fail(e, "infonce launch");
}

int evdr_infonce_distill_fwd_bwd_ws(const float* score_s, const float* score_t, int64_t b, int64_t n, float temperature,
                                    float* loss, float* dscore_or_null, void* workspace, void* hip_stream) {
    if (b < 0 || n < 0) return fail(EVDR_ERR_ARG, "negative size");
    if (!(temperature > 0.f)) return fail(EVDR_ERR_ARG, "temperature must be > 0");
    if (b == 0 || n == 0) return fail(EVDR_ERR_SHAPE, "empty score matrix (b=%lld n=%lld)", (long long)b, (long long)n);
    if (!score_s || !score_t || !loss || !workspace) return fail(EVDR_ERR_ARG, "evdr_infonce_distill_fwd_bwd_ws: null pointer");
    hipError_t e = evdr_launch_infonce_ws(score_s, score_t, b, n, temperature, loss, dscore_or_null, (float*)workspace,
                                          (hipStream_t)hip_stream);
    return e == hipSuccess ? EVDR_OK : hip_fail(e, "infonce launch");
}

}  // extern "C"
