// Shared declarations for the gfx950 kernels and the C-ABI dispatch (include/evdr.h).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/evdr.h"

#define EVDR_VERSION_NUM 303   /* 0.3.3: evdr_maxsim_fwd_prepared_subset + evdr_qcache_* (0.3.2: evdr_adamw_step, AdamW hyper-parameters as doubles; 0.3.1: evdr_maxsim_bwd_adamw_planes; 0.3.0: debug hooks instead of environment switches) */

#define EVDR_D 128              /* embedding width the kernels are specialised for */
#define EVDR_TILE_PATCHES 32    /* patches per LDS tile (two 16-row MFMA halves) */

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// fp32 -> fp16 hi/lo planes: the tensor is scaled by 2^k so that its absmax lands in [2^14, 2^15) (top of the fp16 range:
// the lo plane of every element that matters stays a NORMAL fp16 number).  k from the absmax's exponent field; tensors
// whose absmax is 0, denormal, inf or NaN are left unscaled.
__host__ __device__ static inline int evdr_h2_shift(uint32_t amax_bits) {
    const int e = (int)((amax_bits >> 23) & 0xFFu);
    return (e == 0 || e == 255) ? 0 : 141 - e;
}

// ---- forward --------------------------------------------------------------------------------
struct EvdrFwdParams {
    const uint16_t* Q;          // bf16 planes; query q, token n at Q + q*q_stride + n*128
    int64_t q_stride;           // elements between queries
    int64_t q_plane_stride;     // elements between Q planes
    const uint16_t* P;          // bf16 planes; page-major (np, lp, 128) per plane
    int64_t p_stride;           // elements between pages
    int64_t p_plane_stride;     // elements between P planes
    const uint32_t* q_amax;     // fp16 hi/lo planes (nplanes = 2): absmax bits of the fp32 tensor they were split from
    const uint32_t* p_amax;     //   (evdr_h2_shift gives the power-of-two the planes were scaled by); null = unscaled
    const uint8_t* qmask;       // (nq, lq) or null
    const int32_t* qlist;       // later token slices: compacted indices of the queries with a valid token in the slice, or null
    const int32_t* qcount;      //   ... and their number (device memory)
    const uint32_t* tilemask;   // (np, ntiles)
    const uint32_t* pageflags;  // (np)
    float* out;                 // (nq, out_stride)
    int64_t out_stride;
    uint16_t* argmax;           // (nq, np, lq) or null
    int nq, lq, np, lp, ntiles; // lq = tokens scored by THIS launch (<= 32), starting at token tok0
    int tok0, lq_total;         // lq_total = row length of qmask / argmax (queries longer than 32 tokens
    int accumulate;             //   are scored in 32-token slices, later slices add into out)
    int pages_per_block, n_qgroups, n_chunks;
    int64_t per_token;          // > 0: Q holds this many SINGLE-token queries packed 32 to a "query"; out / argmax get one
                                //      row per token (out (per_token, np), argmax (per_token, np)) instead of the token sum
    int inblock_refill;         // staged kernel: issue the ring refill inside the MFMA block (else right after the barrier)
    unsigned long long* dbg;    // diagnostic builds only: per-wave cycle sums (null in production)
};

// Pages per workgroup.  Workgroups are equal-sized and one fits per CU, so a launch runs in ceil(workgroups / 256) rounds
// of `ppb` pages each plus a fixed cost per workgroup (query load + ring fill, worth about 16 tiles of MFMA work).
// Pick the ppb in [lo, 64] that minimises rounds x (ppb + fixed): this avoids e.g. 1539 workgroups = 6 full rounds + a
// seventh for 3 workgroups.  lo: a workgroup should stream >= ~64 tiles, otherwise the fixed cost dominates -- unless
// that would leave CUs without any workgroup.  Many rounds (>= 64) need no tuning: take 64 pages.
static inline int evdr_pages_per_block(int64_t np, int64_t n_qgroups, int64_t ntiles, int64_t slots = 256) {
    const int64_t total = np * n_qgroups;
    int64_t lo = (64 + ntiles - 1) / ntiles;
    const int64_t fill = (total + slots - 1) / slots;    // pages per workgroup that still gives every workgroup slot one workgroup
    if (lo > fill) lo = fill;
    if (lo < 1) lo = 1;
    int64_t hi = total / (6 * slots);                    // ~6 rounds
    if (hi > 64) hi = 64;
    if (hi < lo) hi = lo;
    if (hi > np) hi = np;
    if (lo > hi) lo = hi;
    if (total / (slots * hi) >= 64) return (int)hi;      // tail round <= 1.5 % whatever the choice
    int64_t best = hi;
    double best_cost = 1e30;
    for (int64_t ppb = np < 64 ? np : 64; ppb >= lo; --ppb) {
        const int64_t wgs = ((np + ppb - 1) / ppb) * n_qgroups;
        const int64_t rounds = (wgs + slots - 1) / slots;
        const double cost = (double)rounds * ((double)ppb + 16.0 / (double)ntiles);
        if (cost < best_cost) { best_cost = cost; best = ppb; }   // ties: the larger ppb (fewer workgroups)
    }
    return (int)best;
}

// The dynamic-LDS limit of a kernel is a per-(function, device) attribute: remember the devices it was raised on, so that
// one process driving several GPUs (torch.cuda.set_device between calls) gets it on each of them.
// Safe from several host threads: the bit set is atomic, and two threads that both find the bit clear both raise the limit
// to the same value (idempotent) before either launches.
static inline hipError_t evdr_ensure_dyn_lds(const void* kern, int bytes, std::atomic<uint64_t>& devs_done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (devs_done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) devs_done.fetch_or(bit, std::memory_order_release);
    return e;
}

// debug hooks (maxsim_fwd.hip): forced kernel variant, pages-per-workgroup override, name of the last dispatched instance
int evdr_fwd_variant_exchange(int v);
int evdr_pages_per_block_exchange(int v);
int evdr_pages_per_block_override();
void evdr_note_fwd_kernel(const char* name);
const char* evdr_last_fwd_kernel_name();

// launches (defined in the .hip files; all enqueue on `stream` and return the launch status)
hipError_t evdr_launch_maxsim_fwd(const EvdrFwdParams& p, int nplanes, bool want_argmax, hipStream_t stream);
hipError_t evdr_launch_maxsim_fwd16(const EvdrFwdParams& p, int qw, int waves, int nplanes, bool want_argmax, int geom,
                                    hipStream_t stream);
hipError_t evdr_launch_pack_pmask(const uint8_t* pmask, int64_t np, int64_t lp, uint32_t* tilemask,
                                  uint32_t* pageflags, hipStream_t stream);
hipError_t evdr_launch_build_qlist(const uint8_t* qmask, int64_t nq, int64_t lq, int64_t tok0, int32_t* qlist, int32_t* qcount,
                                   hipStream_t stream);
hipError_t evdr_launch_split_f32(const float* x, int64_t rows, uint16_t* planes, uint32_t* amax_bits, hipStream_t stream);
hipError_t evdr_launch_split_f32_segments(const float* x, int64_t rows, int64_t seg_rows, uint16_t* planes, uint32_t* amax_bits,
                                          hipStream_t stream);
hipError_t evdr_launch_split_f32_pages(const float* x, int64_t rows, uint16_t* planes, uint32_t* amax_bits, const uint8_t* rowmask,
                                       int64_t rows_per_page, uint32_t* pageflags, hipStream_t stream);
hipError_t evdr_launch_flag_nonfinite(const void* P, int kind, const uint8_t* pmask, int64_t np, int64_t lp, int64_t p_stride,
                                      uint32_t* pageflags, hipStream_t stream);
hipError_t evdr_launch_maxsim_bwd(const float* g, const float* Q, const uint8_t* qmask, const uint8_t* pmask,
                                  const uint16_t* argmax, float* dP, int64_t nq, int64_t lq, int64_t np,
                                  int64_t lp, hipStream_t stream);
hipError_t evdr_launch_maxsim_bwd_adamw(const float* g, const float* Q, const uint8_t* qmask, const uint8_t* pmask,
                                        const uint16_t* argmax, float* x, float* exp_avg, float* exp_avg_sq, int64_t nq,
                                        int64_t lq, int64_t np, int64_t lp, double lr, double beta1, double beta2, double eps,
                                        double weight_decay, double bc1, double bc2_sqrt, float eps_norm, const void* state,
                                        void* next_planes, uint32_t* next_amax, uint32_t* pageflags, hipStream_t stream);
hipError_t evdr_launch_adamw_advance(void* state, double beta1, double beta2, hipStream_t stream);
hipError_t evdr_launch_adamw(const float* g, float* x, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                             double beta2, double eps, double weight_decay, double bc1, double bc2_sqrt, hipStream_t stream);
hipError_t evdr_launch_maxsim_bwd_q(const float* g, const float* P, const uint8_t* qmask, const uint32_t* pageflags,
                                    const uint16_t* argmax, float* dQ, float* partials, int64_t nq, int64_t lq, int64_t np,
                                    int64_t lp, hipStream_t stream);
int evdr_bwd_q_segments(int64_t pairs, int64_t np);
hipError_t evdr_launch_l2norm_fwd(const float* x, const uint8_t* rowmask, int64_t rows, float eps, float* y, float* norm,
                                  uint16_t* planes, uint32_t* amax_bits, uint32_t* pageflags, int64_t rows_per_page,
                                  hipStream_t stream);
hipError_t evdr_launch_l2norm_bwd(const float* gy, const float* x, const uint8_t* rowmask, const float* norm, int64_t rows,
                                  float eps, float* dx, hipStream_t stream);
hipError_t evdr_launch_topk(const float* scores, const int32_t* idx_map, int64_t nq, int64_t n,
                            int64_t row_stride, int32_t idx_base, int k, float* top_scores, int32_t* top_idx,
                            void* workspace, hipStream_t stream);
int evdr_topk_segments(int64_t nq, int64_t n);
hipError_t evdr_launch_qcache_lookup_plan(const EvdrQCache& c, const void* Q, const uint8_t* qmask, const uint32_t* q_amax, int64_t nq,
                                          uint64_t* hashes, int32_t* hit, int32_t* qsel, int32_t* qsel_count, uint32_t* ticket,
                                          hipStream_t stream);
hipError_t evdr_launch_qcache_exchange(const EvdrQCache& c, const void* Q, const uint8_t* qmask, int64_t nq, const int32_t* hit,
                                       float* out, int64_t out_stride, hipStream_t stream);
hipError_t evdr_launch_infonce(const float* ss, const float* st, int64_t b, int64_t n, float temperature,
                               float* loss, float* dscore, float* row_loss, hipStream_t stream);
hipError_t evdr_launch_infonce_ws(const float* ss, const float* st, int64_t b, int64_t n, float temperature, float* loss,
                                  float* dscore, float* workspace, hipStream_t stream);
