// Torch-free user of the round-6 entry points (include/evdr.h): HIP runtime + libevdr.so only.
//   evdr_pack_pmask + evdr_flag_nonfinite + evdr_maxsim_fwd_prepared        a resident bf16 corpus, scored once as the reference
//   evdr_maxsim_fwd_prepared_subset                                         a device-side list of queries: only those rows are written
//   evdr_qcache_workspace + evdr_maxsim_fwd_prepared_cached                 the score-row cache of frozen pages: three batches
//       (all misses; all hits; half known) must each equal evdr_maxsim_fwd_prepared bit for bit, and the device-side miss count must
//       be 8 / 0 / 4 -- with no host synchronisation between the calls of a batch.
// The reference re-scores its frozen teacher every step (mainv2_iter_distill_infonce.py:282-283).  Built and run by
// tests/test_gpu_cabi.py; prints "cabi_qcache OK".
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "evdr.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define EV(x) do { int rc_ = (x); if (rc_ != EVDR_OK) { printf("evdr status %d: %s (%s:%d)\n", rc_, evdr_last_error(), __FILE__, __LINE__); return 3; } } while (0)

static uint16_t to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

template <typename T>
static T* dev_alloc(size_t n, bool zero = false) {
    void* p = nullptr;
    if (hipMalloc(&p, n * sizeof(T) + 16) != hipSuccess) return nullptr;
    if (zero && hipMemset(p, 0, n * sizeof(T)) != hipSuccess) return nullptr;
    return (T*)p;
}

int main() {
    const int64_t nq = 8, lq = 32, np = 37, lp = 206, d = 128, pool = 12;
    std::vector<uint16_t> Qpool(pool * lq * d), P(np * lp * d);
    std::vector<uint8_t> pm(np * lp);
    srand(11);
    auto rnd = [] { return (float)rand() / (float)RAND_MAX - 0.5f; };
    for (auto& v : Qpool) v = to_bf16(rnd() * 0.2f);
    for (auto& v : P) v = to_bf16(rnd() * 0.2f);
    for (auto& v : pm) v = (rand() % 5) != 0;
    for (int m = 0; m < lp; ++m) pm[4 * lp + m] = 0;                    // a page without a valid patch

    hipStream_t st;
    CK(hipStreamCreate(&st));
    uint16_t* dP = dev_alloc<uint16_t>(P.size());
    uint16_t* dQ = dev_alloc<uint16_t>(nq * lq * d);
    uint8_t* dpm = dev_alloc<uint8_t>(pm.size());
    const int64_t ntiles = (lp + 31) / 32;
    uint32_t* tilemask = dev_alloc<uint32_t>(np * ntiles);
    uint32_t* pageflags = dev_alloc<uint32_t>(np);
    float* out_ref = dev_alloc<float>(nq * np);
    float* out = dev_alloc<float>(nq * np);
    int32_t* qsel = dev_alloc<int32_t>(nq);
    int32_t* qcount = dev_alloc<int32_t>(1);
    CK(hipMemcpy(dP, P.data(), P.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dpm, pm.data(), pm.size(), hipMemcpyHostToDevice));
    EV(evdr_pack_pmask(dpm, np, lp, tilemask, pageflags, st));
    EV(evdr_flag_nonfinite(dP, EVDR_BF16, dpm, np, lp, lp * d, pageflags, st));

    // the cache: capacity 16 rows of 32 x 128 bf16, table of 32 slots
    EvdrQCache c{};
    const int64_t cap = 16;
    c.capacity = cap; c.n_slots = 32; c.row_bytes = lq * d * 2; c.lq = lq; c.np = np; c.hash_mask = ~0ull;
    c.slots = dev_alloc<int32_t>(c.n_slots, true);
    c.ent_hash = dev_alloc<uint64_t>(cap);
    c.ent_k = dev_alloc<int32_t>(cap);
    c.ent_q = dev_alloc<uint8_t>(cap * c.row_bytes);
    c.ent_mask = dev_alloc<uint8_t>(cap * lq);
    c.ent_scores = dev_alloc<float>(cap * np);
    c.n_entries = dev_alloc<int32_t>(1, true);
    const size_t wsb = evdr_qcache_workspace(nq);
    uint8_t* ws = dev_alloc<uint8_t>(wsb, true);                        // zeroed ONCE: it carries the ticket word between calls
    if (!c.slots || !c.ent_q || !ws || wsb < (size_t)nq * 16) { printf("allocation failed\n"); return 2; }

    std::vector<float> a(nq * np), b(nq * np);
    const int first_query[3] = {0, 0, 4};                               // batches: pool rows 0..7, 0..7 again, 4..11
    const int want_misses[3] = {8, 0, 4};
    for (int batch = 0; batch < 3; ++batch) {
        CK(hipMemcpyAsync(dQ, Qpool.data() + (size_t)first_query[batch] * lq * d, nq * lq * d * 2, hipMemcpyHostToDevice, st));
        EV(evdr_maxsim_fwd_prepared(dQ, dP, nullptr, tilemask, pageflags, out_ref, np, nullptr, nq, lq, np, lp, 1, lp * d, 0, nullptr, nullptr, nullptr, st));
        CK(hipMemsetAsync(out, 0xFF, nq * np * 4, st));
        EV(evdr_maxsim_fwd_prepared_cached(&c, dQ, dQ, dP, nullptr, tilemask, pageflags, out, np, nq, lp, 1, lp * d, 0, nullptr, nullptr, ws, wsb, st));
        CK(hipMemcpyAsync(a.data(), out_ref, a.size() * 4, hipMemcpyDeviceToHost, st));
        CK(hipMemcpyAsync(b.data(), out, b.size() * 4, hipMemcpyDeviceToHost, st));
        int32_t misses = -1;
        CK(hipMemcpyAsync(&misses, ws + wsb - 256, 4, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        if (memcmp(a.data(), b.data(), a.size() * 4) != 0) { printf("batch %d: cached scores differ from the plain forward\n", batch); return 1; }
        if (misses != want_misses[batch]) { printf("batch %d: %d queries scored, expected %d\n", batch, misses, want_misses[batch]); return 1; }
        for (int q = 0; q < nq; ++q)
            if (a[q * np + 4] != 0.0f) { printf("all-masked page scored %g\n", a[q * np + 4]); return 1; }
    }
    int32_t entries = -1;
    CK(hipMemcpy(&entries, c.n_entries, 4, hipMemcpyDeviceToHost));
    if (entries != 12) { printf("entries %d, expected 12\n", entries); return 1; }

    // subset forward on its own: rows 1, 2, 6 of the last batch, everything else untouched
    const int32_t sel[8] = {1, 2, 6, 0, 0, 0, 0, 0}, three = 3;
    CK(hipMemcpyAsync(qsel, sel, sizeof(sel), hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(qcount, &three, 4, hipMemcpyHostToDevice, st));
    CK(hipMemsetAsync(out, 0, nq * np * 4, st));
    EV(evdr_maxsim_fwd_prepared_subset(dQ, dP, nullptr, tilemask, pageflags, out, np, nq, lq, np, lp, 1, lp * d, 0, nullptr, nullptr, qsel, qcount, st));
    CK(hipMemcpyAsync(b.data(), out, b.size() * 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    for (int q = 0; q < nq; ++q) {
        const bool listed = q == 1 || q == 2 || q == 6;
        for (int p = 0; p < np; ++p) {
            const float want = listed ? a[q * np + p] : 0.0f;
            if (memcmp(&want, &b[q * np + p], 4) != 0) { printf("subset forward: row %d page %d is %g, expected %g\n", q, p, b[q * np + p], want); return 1; }
        }
    }
    // error paths stay status codes
    if (evdr_maxsim_fwd_prepared_cached(&c, dQ, dQ, dP, nullptr, tilemask, pageflags, out, np, nq, lp, 1, lp * d, 0, nullptr, nullptr, ws, 8, st) != EVDR_ERR_WORKSPACE) {
        printf("a short workspace was accepted\n");
        return 1;
    }
    printf("cabi_qcache OK\n");
    return 0;
}
