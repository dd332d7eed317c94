/*
 * evdr.h -- C ABI of the MI355X-native late-interaction (MaxSim) scorer.
 *
 * The reference (kimjy-st/Efficient-Visual-Document-Retrieval) has no FFI layer: its boundary
 * for this path is a set of Python function signatures in evaluator/retrieval.py.  Each entry
 * point below names the reference interface it replaces (paths relative to the reference root).
 * The Python host shim (efficient-visual-document-retrieval_amd/evaluator/retrieval.py) binds these with
 * ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless it says "host";
 *   - plain pointers and sizes only, no framework types; all functions return an int status
 *     (EVDR_OK == 0) and never throw/abort across the ABI; evdr_last_error() gives the text;
 *   - nothing here allocates, frees or synchronises: scratch memory comes in through
 *     `workspace` (query the size first), work is enqueued on `hip_stream` (a hipStream_t, may
 *     be NULL for the default stream) and the call returns immediately (graph-capture safe);
 *   - THREADING: entry points may be called from several host threads at once, each call on whatever stream it names (the
 *     reference itself drives its scorer from a single Python thread, DataLoader(num_workers=0);
 *     mainv2_iter_distill_infonce.py:269-321).  The library keeps no mutable state between calls except (a) per kernel
 *     instance, the set of devices on which its dynamic-LDS limit has been raised -- an atomic bit set, and raising the limit
 *     twice is harmless.  evdr_last_error(), evdr_last_fwd_kernel() and the two debug overrides at the end of this file are
 *     THREAD-LOCAL: a thread reads and sets its own, no caller can change what another caller's launch does.  Ordering between calls that
 *     touch the same buffers is the caller's (streams / events), as with any HIP launch; workspaces are per call.
 *     tests/cabi/cabi_threads.cpp: eight threads, one kernel family each, first launches racing, bit-equal to serial calls.
 *     Several processes per node (one per GPU: corpus.py / bench.py) are the multi-GPU form;
 *   - D (embedding width) is 128 (ColPali / ColQwen projection width) for every entry point that takes `d`; wider
 *     embeddings (up to 256) are scored through evdr_maxsim_fwd_prepared with nplanes = 4, and their gradients -- linear in the
 *     columns -- by one evdr_maxsim_bwd / evdr_maxsim_bwd_q call per 128-column block (what the Python host shim does);
 *   - masks are one byte per token, 0 = masked (torch.bool storage);
 *   - dtype: EVDR_F32 inputs are scored to fp32 accuracy (fp16 hi/lo planes of the power-of-two-scaled
 *     tensors, 3 MFMA products, error below the rounding noise of an fp32 accumulation); EVDR_BF16
 *     inputs are used as they are (products exact in fp32).
 *
 * Non-finite inputs.  For finite inputs every result below is the reference's to the stated tolerance.  A NaN is handled
 * where the reference's NaN lands (torch.max propagates it, evaluator/retrieval.py:201; the hardware max does not, so the
 * inputs are inspected instead):
 *   - a NaN in a VALID patch of page p (masked patches are replaced by -1e4 before the max and do not count)
 *     -> out[q][p] = NaN for every q;
 *   - a NaN in any token of query q, masked tokens included (NaN * 0 = NaN, :207) -> out[q][p] = NaN for every page p
 *     that has a valid patch (an all-masked page keeps its exact 0);
 *   - +-Inf elements are treated like NaN (torch yields +-Inf or NaN there depending on signs: the one divergence).
 * Pages are inspected when they are prepared: evdr_maxsim_fwd does it per call; for a resident corpus call
 * evdr_flag_nonfinite once after evdr_pack_pmask (evdr_l2norm_fwd_split can report its rows on the way).  Queries are
 * inspected inside the kernel.  argmax entries of NaN pairs are unspecified; evdr_topk ranks NaN first (like torch.topk);
 * the gradient kernels propagate whatever non-finite upstream gradient they are given, positions unspecified.
 * Limitation: for queries longer than 32 tokens, a non-finite element in a MASKED token beyond the first 32 goes unseen
 * if the query has no valid token in that 32-token slice.
 */
#ifndef EVDR_H
#define EVDR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The entry points below are the library's ONLY dynamic symbols: libevdr.so is built with -fvisibility=hidden and each of them
 * carries default visibility (tests/test_cabi_exports.py compares `nm -D --defined-only` with this header, both ways). */
#define EVDR_API __attribute__((visibility("default")))

#define EVDR_OK            0
#define EVDR_ERR_ARG       1   /* null pointer / bad enum / negative size            */
#define EVDR_ERR_SHAPE     2   /* unsupported shape (D != 128, Lp > 65535 w/ argmax) */
#define EVDR_ERR_WORKSPACE 3   /* workspace missing or too small                     */
#define EVDR_ERR_HIP       4   /* a HIP runtime call or kernel launch failed         */

#define EVDR_F32  0
#define EVDR_BF16 1
#define EVDR_F16  2            /* only evdr_flag_nonfinite: the hi plane of fp16 hi/lo planes */

#define EVDR_TOPK_MAX 128      /* k_values max is 100 (evaluator/retrieval.py:223)   */

EVDR_API int         evdr_version(void);          /* 10000*major + 100*minor + patch */
EVDR_API const char* evdr_last_error(void);       /* host string, thread-local, valid until the next call */

/* ---- corpus preparation (the steps either side of the hot loop; cacheable for a static corpus) ---- */

/* Pack a (np, lp) byte mask into per-32-patch tile words + per-page flags.
 * tilemask: np * ceil(lp/32) uint32 (bit m of word t = pmask[p][32 t + m]);
 * pageflags: np uint32:
 *   bit0 = page has a valid patch  [doc_has_token, evaluator/retrieval.py:192]
 *   bit1 = page has a masked patch [the -1e4 fill takes part in the max, :198]
 *   bit2 = the valid patches are exactly ONE range [va, vb) -- the whole page, a ragged prefix, an image between masked
 *          text tokens (what utils/preprocess_data.py:101 produces), or no patch at all: then va = bits 4..15 (< 4096),
 *          vb = bits 16..31, and the kernels walk / fetch only that range; otherwise (holes) bits 16..31 = index of the
 *          first masked patch and the tile words decide
 *   bit3 = a valid patch holds a NaN / Inf element (set by evdr_flag_nonfinite, never here). */
EVDR_API int evdr_pack_pmask(const uint8_t* pmask, int64_t np, int64_t lp,
                    uint32_t* tilemask, uint32_t* pageflags, void* hip_stream);

/* Split `rows` x 128 fp32 into two fp16 planes hi/lo of x * 2^k, hi + lo == x * 2^k to 2^-22 relative.  k is one
 * power of two per tensor, chosen from its absmax so that the scaled absmax lies in [2^14, 2^15) (no fp16 overflow, and
 * the lo plane of every element that matters is a normal number).  planes: 2 * rows * 128 uint16 (fp16 bits),
 * plane-major.  amax_bits: 1 uint32 on the device = bits of max|x| (the kernels derive k from it; keep it with the
 * planes).  Three fp16 MFMA products lo*hi + hi*lo + hi*hi then give the fp32 dot product to below fp32 rounding noise. */
EVDR_API int evdr_split_f32(const float* x, int64_t rows, uint16_t* planes, uint32_t* amax_bits, void* hip_stream);

/* evdr_split_f32 of MANY small tensors laid end to end, in one launch: segment s = rows [s * seg_rows, (s + 1) * seg_rows) of
 * x (the last segment may be short; seg_rows <= 2048, at most 65535 segments) -- the batches of a training epoch, whose
 * queries are known when the epoch starts (the reference draws them per step from a DataLoader,
 * mainv2_iter_distill_infonce.py:81).  Every segment keeps ITS OWN absmax word amax_bits[s] and power of two: its planes
 * are bit for bit those of evdr_split_f32 applied to that segment alone.  Segment s's planes are the contiguous
 * (2, rows_s, 128) block at planes + s * 2 * seg_rows * 128 -- what evdr_maxsim_fwd_prepared takes as Q planes.
 * planes: 2 * rows * 128 uint16 (rounded up to whole segments: ceil(rows / seg_rows) * 2 * seg_rows * 128). */
EVDR_API int evdr_split_f32_segments(const float* x, int64_t rows, int64_t seg_rows, uint16_t* planes, uint32_t* amax_bits, void* hip_stream);

/* Report non-finite page content: sets bit 3 of pageflags[p] (made by evdr_pack_pmask) when a valid patch of page p holds a
 * NaN or +-Inf element; the forward kernels then return NaN for that page ("Non-finite inputs" above).  P: (np, lp, 128)
 * of dtype EVDR_F32, EVDR_BF16 or EVDR_F16 (the hi plane of fp16 hi/lo planes), `p_stride` elements between pages.  One
 * read of P; flags are only ever set, re-run evdr_pack_pmask to clear them. */
EVDR_API int evdr_flag_nonfinite(const void* P, int dtype, const uint8_t* pmask, int64_t np, int64_t lp, int64_t p_stride,
                        uint32_t* pageflags, void* hip_stream);

/* ---- A1: score_multi_vector_masked (evaluator/retrieval.py:166-213) ---------------------------------
 * out[q,p] = sum_n qmask[q,n] * has(p) * max_m( Q[q,n,:]·P[p,m,:] if pmask[p,m] else -1e4 )
 * Q (nq,lq,128), P (np,lp,128) of `dtype`, row-major, contiguous in the last two dims;
 * strides_or_null = {q_stride, p_stride} in ELEMENTS between consecutive queries / pages
 * (NULL = dense).  qmask (nq,lq) / pmask (np,lp) bytes; either may be NULL = all valid.
 * out (nq,np) fp32 dense.  argmax_or_null: (nq,np,lq) uint16, first maximal patch index per
 * query token (what torch.max picks, :201) -- needed only for evdr_maxsim_bwd.
 * chunk_p of the reference is a memory knob with no numerical effect and has no counterpart.
 * lq == 1 with dense queries (the single-token "virtual queries" of mainv3_iter_liscore_QA_hardtoken.py:428-434) is
 * scored 32 queries to an MFMA tile internally; shapes and results are those of the general case.
 * Cost note: this entry PREPARES the pages on every call -- it packs pmask, scans P for NaN / Inf (one extra read of
 * all of P: with 1-8 queries per call, where the scorer itself only streams P once, that roughly doubles the call) and,
 * for EVDR_F32, splits P into fp16 planes.  A caller that scores the same pages more than once should prepare them once
 * (evdr_pack_pmask + evdr_flag_nonfinite [+ evdr_split_f32]) and call evdr_maxsim_fwd_prepared. */
EVDR_API size_t evdr_maxsim_fwd_workspace(int64_t nq, int64_t lq, int64_t np, int64_t lp, int dtype);
EVDR_API int evdr_maxsim_fwd(const void* Q, const void* P, const uint8_t* qmask, const uint8_t* pmask,
                    float* out, uint16_t* argmax_or_null,
                    int64_t nq, int64_t lq, int64_t np, int64_t lp, int64_t d, int dtype,
                    const int64_t* strides_or_null,
                    void* workspace, size_t workspace_bytes, void* hip_stream);

/* Same computation on a PREPARED (resident) corpus: planes + packed masks made once with
 * evdr_split_f32 / evdr_pack_pmask.  nplanes = 1 (bf16 tensors as they are), 2 (fp16 hi/lo planes of fp32
 * tensors, with the absmax words evdr_split_f32 produced for Q and P; NULL = planes are unscaled) or 4 (embeddings of
 * width 256: the reference takes any width, evaluator/retrieval.py:173 -- the two 128-column blocks of each tensor as fp16 hi/lo
 * planes under ONE absmax word per tensor, plane index = 2 * (0 hi | 1 lo) + column block, i.e. evdr_split_f32 of the tensor
 * rearranged to (column block, row, 128); widths of 129..255 ride on zero columns; one query per wave, six plane products per
 * k-step into one accumulator chain, same accuracy and NaN rules as nplanes = 2; evdr_flag_nonfinite once per hi plane).  Q planes are
 * (nplanes, nq, lq, 128); P planes are nplanes slabs `p_plane_stride` elements apart, each
 * (np, lp, 128) with `p_stride` elements between pages.  out row stride = out_stride floats, so a
 * shard can write its column block of a wider (nq, N) matrix.  This is the bench / retrieval
 * hot path (SURVEY §8(d),(e)).  qlist_ws_or_null: (nq + 1) int32 of scratch, used when lq > 32 to score the later
 * 32-token slices only for the queries that have valid tokens there (query sets are padded to their longest member). */
EVDR_API int evdr_maxsim_fwd_prepared(const uint16_t* Qplanes, const uint16_t* Pplanes,
                             const uint8_t* qmask, const uint32_t* tilemask, const uint32_t* pageflags,
                             float* out, int64_t out_stride, uint16_t* argmax_or_null,
                             int64_t nq, int64_t lq, int64_t np, int64_t lp,
                             int nplanes, int64_t p_stride, int64_t p_plane_stride,
                             const uint32_t* q_amax_or_null, const uint32_t* p_amax_or_null,
                             int32_t* qlist_ws_or_null, void* hip_stream);

/* The same forward for a device-side SUBSET of the queries: only the queries qsel[0 .. *qsel_count) (ascending indices into the
 * nq rows of Qplanes, both in device memory) are scored and only their rows of `out` are written; workgroups whose queries all lie
 * beyond *qsel_count leave at once.  The count is read on the device: the host needs no synchronisation to know which queries a
 * preceding kernel selected (the score-row cache below).  lq <= 32 and lq != 1-packed layouts only (EVDR_ERR_SHAPE otherwise); no argmax. */
EVDR_API int evdr_maxsim_fwd_prepared_subset(const uint16_t* Qplanes, const uint16_t* Pplanes,
                             const uint8_t* qmask, const uint32_t* tilemask, const uint32_t* pageflags,
                             float* out, int64_t out_stride,
                             int64_t nq, int64_t lq, int64_t np, int64_t lp,
                             int nplanes, int64_t p_stride, int64_t p_plane_stride,
                             const uint32_t* q_amax_or_null, const uint32_t* p_amax_or_null,
                             const int32_t* qsel, const int32_t* qsel_count, void* hip_stream);

/* ---- score-row cache of a FROZEN page tensor (mainv2_iter_distill_infonce.py:282-283) --------------------------------------------
 * The reference's step re-scores the frozen teacher pages for every batch, every epoch, although a query's teacher scores never
 * change ("caching them is a legal, result-identical optimisation").  evdr_maxsim_fwd_prepared_cached is
 * evdr_maxsim_fwd_prepared (no argmax) through a (query row -> score row) cache that lives on the device, for ONE prepared page
 * tensor, without any host synchronisation.  Three launches on `hip_stream`:
 *   lookup + plan   one workgroup per query: 64-bit hash of (query row bits, mask row, plane shift k of the batch), probe of an
 *                   open-addressing table, and -- on a hash match -- a comparison of the FULL stored row and mask, so that a hash
 *                   collision is a miss.  The last workgroup to finish turns the misses, in ascending order, into the query list of
 *                   the forward and, while the cache has room, gives each a fresh entry and enters it into the table;
 *   forward         evdr_maxsim_fwd_prepared_subset over that device-side list (all hits: every workgroup leaves at once);
 *   exchange        rows of hits are copied out of the cache into `out`, rows of stored misses (score row, query row, mask) into it.
 * The plane shift k (evdr_split_f32 scales a batch by one power of two taken from ITS absmax) is part of the key: a query that comes
 * back in a batch with another absmax exponent is scored again, so a cached row always equals, bit for bit, the row
 * evdr_maxsim_fwd_prepared would write for this very call.  Nothing is evicted: a full cache keeps its entries and later misses are
 * just scored.  One cache belongs to one stream order of calls (never two streams at once).
 * Qrows: the batch as the caller holds it, (nq, cache->lq, 128) dense, cache->row_bytes per query (fp32 or bf16) -- its bits are the
 * key; Qplanes / q_amax: the same batch as the forward takes it.  workspace: evdr_qcache_workspace(nq) bytes, ZEROED by the caller
 * before its first use and then left to the library (it carries a ticket word between calls; after a call the int32 at byte offset
 * evdr_qcache_workspace(nq) - 256 holds the number of queries that had to be scored).
 * The struct is plain host memory; all of its pointers are device memory owned by the caller. */
typedef struct EvdrQCache {
    int32_t*  slots;        /* (n_slots) open-addressing table: 0 = empty, else entry index + 1; zero-initialised by the caller */
    int64_t   n_slots;      /* a power of two, >= 2 * capacity                                                                      */
    uint64_t* ent_hash;     /* (capacity)                                                                                         */
    int32_t*  ent_k;        /* (capacity) plane shift the entry was scored with                                                   */
    uint8_t*  ent_q;        /* (capacity, row_bytes) query rows as the caller's dtype holds them                                  */
    uint8_t*  ent_mask;     /* (capacity, lq) mask rows, 0 / 1                                                                     */
    float*    ent_scores;   /* (capacity, np)                                                                                     */
    int32_t*  n_entries;    /* (1) entries handed out so far; zero-initialised by the caller                                      */
    int64_t   capacity, row_bytes, lq, np;
    uint64_t  hash_mask;    /* all ones; tests narrow it (0 = every row collides) to exercise the full-row comparison             */
} EvdrQCache;
EVDR_API size_t evdr_qcache_workspace(int64_t nq);
EVDR_API int evdr_maxsim_fwd_prepared_cached(const EvdrQCache* cache, const void* Qrows, const uint16_t* Qplanes, const uint16_t* Pplanes,
                             const uint8_t* qmask, const uint32_t* tilemask, const uint32_t* pageflags,
                             float* out, int64_t out_stride, int64_t nq, int64_t lp,
                             int nplanes, int64_t p_stride, int64_t p_plane_stride,
                             const uint32_t* q_amax_or_null, const uint32_t* p_amax_or_null,
                             void* workspace, size_t workspace_bytes, void* hip_stream);

/* ---- A6: autograd of A1 w.r.t. P (loss.backward(), mainv2_iter_distill_infonce.py:290) ---------------
 * dP[p,m,:] = sum_{q,n} g[q,p] * qmask[q,n] * has(p) * [m == argmax[q,p,n]] * Q[q,n,:]
 * g (nq,np) fp32; Q (nq,lq,128) fp32; argmax from evdr_maxsim_fwd; dP (np,lp,128) fp32 is
 * OVERWRITTEN (every element written, masked rows get exact zeros). */
EVDR_API int evdr_maxsim_bwd(const float* g, const float* Q, const uint8_t* qmask, const uint8_t* pmask,
                    const uint16_t* argmax, float* dP,
                    int64_t nq, int64_t lq, int64_t np, int64_t lp, int64_t d, void* hip_stream);

/* ---- A6, query side: autograd of A1 w.r.t. Q (the reference's function is differentiable in both arguments) ---------
 * dQ[q,n,:] = qmask[q,n] * sum_p g[q,p] * has(p) * P[p, argmax[q,p,n], :]
 * g (nq,np) fp32; P (np,lp,128) fp32 dense; argmax from evdr_maxsim_fwd; dQ (nq,lq,128) fp32 is OVERWRITTEN.
 * Deterministic: with few (query, token) pairs the page range is split over workgroups, whose partial sums go to the
 * workspace and are added in a fixed order (no float atomics) -- the same bits run after run. */
EVDR_API size_t evdr_maxsim_bwd_q_workspace(int64_t nq, int64_t lq, int64_t np, int64_t lp);
EVDR_API int evdr_maxsim_bwd_q(const float* g, const float* P, const uint8_t* qmask, const uint8_t* pmask,
                      const uint16_t* argmax, float* dQ,
                      int64_t nq, int64_t lq, int64_t np, int64_t lp, int64_t d,
                      void* workspace, size_t workspace_bytes, void* hip_stream);

/* ---- A6 + A4 backward + the optimizer of A7 in ONE launch (mainv2_iter_distill_infonce.py:279,290-291) ---------------
 * Same gather as evdr_maxsim_bwd, but the gradient w.r.t. the normalised pages stays in LDS and the epilogue applies
 * (1) the backward of  y = m x / (||m x|| + l2_eps)  (l2_normalize(Pbar * pmask), utils/preprocess_data.py:8-9) and
 * (2) torch.optim.AdamW's update (decoupled weight decay, bias correction with `step` = 1 for the first update)
 * in place on the raw parameter x (np,lp,128) and its moments exp_avg / exp_avg_sq (same shape, zero-initialised).
 * pmask doubles as the row mask m.  Result-identical to backward() + AdamW.step() of the reference's step.
 * lr / betas / eps / weight_decay are DOUBLES, as Python hands them to torch: the derived constants (1 - lr * weight_decay,
 * 1 - beta, lr / (1 - beta1^step), ...) are formed in double on the host like torch forms them and only then rounded to fp32
 * (1 - float(0.999) would be off by 1.3e-5 relative). */
EVDR_API int evdr_maxsim_bwd_adamw(const float* g, const float* Q, const uint8_t* qmask, const uint8_t* pmask,
                          const uint16_t* argmax, float* x, float* exp_avg, float* exp_avg_sq,
                          int64_t nq, int64_t lq, int64_t np, int64_t lp, int64_t d,
                          double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                          float l2_eps, const void* adamw_state_or_null, void* hip_stream);
/* The same launch, also emitting the NEXT forward's operands: l2_normalize(m * x_new) of the updated parameter as the fp16
 * hi/lo planes evdr_l2norm_fwd_split would produce from it (next_planes: 2 x np x lp x 128 fp16, next_amax: their absmax
 * word; both or neither), non-finite updated rows reported in pageflags_or_null (np words, bit 3) like there.  A training
 * loop that keeps the planes (mainv2_iter_distill_infonce.py:279 normalises the parameter at the top of EVERY step) saves
 * the separate normalise pass: one read of x and one launch per step. */
EVDR_API int evdr_maxsim_bwd_adamw_planes(const float* g, const float* Q, const uint8_t* qmask, const uint8_t* pmask,
                                 const uint16_t* argmax, float* x, float* exp_avg, float* exp_avg_sq,
                                 int64_t nq, int64_t lq, int64_t np, int64_t lp, int64_t d,
                                 double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                                 float l2_eps, const void* adamw_state_or_null,
                                 void* next_planes_or_null, uint32_t* next_amax_or_null, uint32_t* pageflags_or_null,
                                 void* hip_stream);
/* Device-resident step counter for the calls above, so that a whole training step can be captured in a HIP graph and
 * replayed without a scalar from the host: adamw_state = 16 bytes of device memory {int64 step; float bc1; float bc2_sqrt},
 * zero-initialised.  evdr_adamw_advance does step += 1 and refreshes the bias corrections; evdr_maxsim_bwd_adamw with a
 * non-NULL state reads them from there and ignores its `step` argument. */
EVDR_API int evdr_adamw_advance(void* adamw_state, double beta1, double beta2, void* hip_stream);
/* The optimizer of A7 on its own (utils/utils.py:78-80 -> torch.optim.AdamW at torch's default betas / eps, amsgrad off), for
 * callers that keep the reference's autograd step (loss.backward(); opt.step(), mainv2_iter_distill_infonce.py:290-291): one
 * pass over grad, x, exp_avg, exp_avg_sq (n fp32 elements each, 16-byte aligned, dense), in place; `step` >= 1 is the count
 * INCLUDING this update (bias corrections 1 - beta^step).  torch's default (foreach) form makes eight passes. */
EVDR_API int evdr_adamw_step(const float* grad, float* x, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                    double beta2, double eps, double weight_decay, int64_t step, void* hip_stream);

/* ---- A4: l2_normalize (utils/preprocess_data.py:8-9) fused with the page mask, forward and backward -----------------
 * y[r,:] = m_r * x[r,:] / (||m_r * x[r,:]||_2 + eps), m_r = rowmask[r] != 0 (NULL = all ones); rows x 128 fp32.
 * norm_or_null receives ||m_r x_r|| (needed by the backward).  One call replaces the `Pbar * pmask` multiply and
 * the four ATen kernels of l2_normalize on the training step (mainv2_iter_distill_infonce.py:279).
 * Backward: dx = m * ( gy/(n+eps) - (m x) * ((m x).gy) / (n (n+eps)^2) ), with the norm's subgradient 0 at n = 0
 * (what torch autograd yields for x / (x.norm() + eps)). */
EVDR_API int evdr_l2norm_fwd(const float* x, const uint8_t* rowmask_or_null, int64_t rows, int64_t d, float eps,
                    float* y, float* norm_or_null, void* hip_stream);
EVDR_API int evdr_l2norm_bwd(const float* gy, const float* x, const uint8_t* rowmask_or_null, const float* norm,
                    int64_t rows, int64_t d, float eps, float* dx, void* hip_stream);
/* evdr_l2norm_fwd that also (or only: y_or_null = NULL) emits y in evdr_split_f32's format -- fp16 hi/lo planes
 * (2 * rows * 128 uint16) + the absmax word, here the constant bits of 1.0f since |y| <= 1 -- ready for
 * evdr_maxsim_fwd_prepared(nplanes = 2): the normalised student pages of a training step go to the scorer without an
 * fp32 round trip through HBM (mainv2_iter_distill_infonce.py:279-283).  pageflags_or_null (with rows_per_page = lp): a
 * non-finite unmasked row sets bit 3 of its page's flag word on the way (see evdr_flag_nonfinite). */
EVDR_API int evdr_l2norm_fwd_split(const float* x, const uint8_t* rowmask_or_null, int64_t rows, int64_t d, float eps,
                          float* y_or_null, float* norm_or_null, uint16_t* planes, uint32_t* amax_bits,
                          uint32_t* pageflags_or_null, int64_t rows_per_page, void* hip_stream);

/* ---- A8: top-k per query row, replaces the Nq*N .item() loop (mainv2_iter_distill_infonce.py:311-317)
 * scores (nq, n) fp32 with row stride `row_stride`; idx_map_or_null (nq, n) int32 maps a column to
 * the index to report (used when merging per-shard candidate lists), else column + idx_base.
 * Order: score descending, reported index ascending on ties.  k <= EVDR_TOPK_MAX.
 * top_scores/top_idx (nq, k); rows with n < k are padded with (-inf, -1).
 * workspace_or_null (evdr_topk_workspace bytes; 0 = not needed): lets a few long rows be ranked by many workgroups
 * (per-segment candidates, then a merge) instead of one workgroup per row; same result either way. */
EVDR_API size_t evdr_topk_workspace(int64_t nq, int64_t n, int k);
EVDR_API int evdr_topk(const float* scores, const int32_t* idx_map_or_null, int64_t nq, int64_t n,
              int64_t row_stride, int32_t idx_base, int k,
              float* top_scores, int32_t* top_idx,
              void* workspace_or_null, size_t workspace_bytes, void* hip_stream);

/* A1 + A8 in one call on a prepared corpus: scores land in `workspace` (nq*np floats). */
EVDR_API size_t evdr_maxsim_topk_workspace(int64_t nq, int64_t np);
EVDR_API int evdr_maxsim_topk(const uint16_t* Qplanes, const uint16_t* Pplanes,
                     const uint8_t* qmask, const uint32_t* tilemask, const uint32_t* pageflags,
                     int64_t nq, int64_t lq, int64_t np, int64_t lp,
                     int nplanes, int64_t p_stride, int64_t p_plane_stride,
                     const uint32_t* q_amax_or_null, const uint32_t* p_amax_or_null,
                     int32_t idx_base, int k, float* top_scores, int32_t* top_idx,
                     void* workspace, size_t workspace_bytes, void* hip_stream);

/* ---- A5 (+ its gradient): infonce_distillation_loss (criterion.py:56-68) -----------------------------
 * loss = mean_b CE(score_s[b,:]/temperature, argmax_p score_t[b,:]);
 * dscore[b,p] = (softmax(score_s[b,:]/temperature)[p] - [p == target_b]) / (temperature * B).
 * score_s/score_t (b, n) fp32 dense; loss: 1 float -- device memory, or device-accessible pinned host memory (hipHostMalloc:
 * the one 4-byte store then lands on the host and float(loss) needs no copy launch, only an event behind the call);
 * dscore_or_null (b, n);
 * row_loss: b floats of scratch (device). */
EVDR_API int evdr_infonce_distill_fwd_bwd(const float* score_s, const float* score_t, int64_t b, int64_t n,
                                 float temperature, float* loss, float* dscore_or_null,
                                 float* row_loss, void* hip_stream);

/* The same in ONE launch, for a caller that keeps a workspace across calls: `workspace` = (b + 1) 4-byte words of device
 * memory, the last one a ticket counter that is ZERO before the first call and is left zero by every call (the last
 * workgroup to finish reduces the per-row losses in a fixed order, so the loss is the same bits as above).  One workspace
 * per stream that may run the call concurrently. */
EVDR_API int evdr_infonce_distill_fwd_bwd_ws(const float* score_s, const float* score_t, int64_t b, int64_t n,
                                    float temperature, float* loss, float* dscore_or_null,
                                    void* workspace, void* hip_stream);

/* ---- debug hooks: not part of the drop-in surface (tests and A/B measurements only) -----------------------------------
 * The library reads NO environment variable and takes no pointer from one; everything that changes its behaviour comes in
 * through a call.  Both overrides are THREAD-LOCAL (like evdr_last_error): they apply to the launches the CALLING thread issues
 * after the call and to no other thread's -- no process-wide mutable state (tests/cabi/cabi_threads.cpp: a thread that forces a
 * variant does not change what its neighbours dispatch).
 * evdr_debug_set_fwd_variant: force a forward-kernel family for the calling thread's launches (0 = default dispatch,
 * 1 = flat per-tile ring, 2 = staged kernel without the priority schedule, 10/11 = fp16-plane stage shapes (3 / 4 tiles),
 * 30 = one 8-wave workgroup per CU also for 3-12 queries, 31 = no non-temporal corpus stream, 33 / 34 = non-temporal stream of
 * the fp16-plane forward forced on / off (default: on for >= 128 MiB of planes read by <= 2 query groups), 36 = two queries per wave also
 * for small fp16-plane launches (default: one per wave when 16-query workgroups would number <= 128); all variants compute the same
 * scores -- the test sweep runs each against the oracle); returns the previous value.
 * evdr_debug_set_pages_per_block: pages per workgroup for the calling thread's launches (0 = automatic); returns the previous value.
 * evdr_last_fwd_kernel: host string naming the forward instance the last evdr_maxsim_fwd* / evdr_maxsim_topk call of
 * this thread dispatched (template arguments as in csrc/maxsim_fwd16.hip) -- what bench.py records next to its timing. */
EVDR_API int         evdr_debug_set_fwd_variant(int variant);
EVDR_API int         evdr_debug_set_pages_per_block(int pages);
EVDR_API const char* evdr_last_fwd_kernel(void);

#ifdef __cplusplus
}
#endif
#endif /* EVDR_H */
