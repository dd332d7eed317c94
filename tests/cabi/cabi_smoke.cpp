// Torch-free user of the C ABI (include/evdr.h): HIP runtime + libevdr.so only.  Scores a small bf16 problem through
// evdr_maxsim_fwd and evdr_topk and checks it against a scalar host loop of the reference's formula
// (evaluator/retrieval.py:187-211).  Built and run by tests/test_gpu_cabi.py; prints "cabi_smoke OK".
#include <hip/hip_runtime.h>

#include <cmath>
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
static float from_bf16(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int main() {
    const int64_t nq = 5, lq = 9, np = 13, lp = 75, d = 128;
    const int k = 4;
    std::vector<uint16_t> Q(nq * lq * d), P(np * lp * d);
    std::vector<uint8_t> qm(nq * lq), pm(np * lp);
    srand(7);
    auto rnd = [] { return (float)rand() / RAND_MAX - 0.5f; };
    for (auto& v : Q) v = to_bf16(rnd() * 0.2f);
    for (auto& v : P) v = to_bf16(rnd() * 0.2f);
    for (auto& v : qm) v = (rand() % 5) != 0;
    for (auto& v : pm) v = (rand() % 4) != 0;
    for (int m = 0; m < lp; ++m) pm[2 * lp + m] = 0;                    // an all-masked page scores exactly 0

    // host reference
    std::vector<float> want(nq * np);
    for (int q = 0; q < nq; ++q)
        for (int p = 0; p < np; ++p) {
            bool has = false;
            for (int m = 0; m < lp; ++m) has |= pm[p * lp + m] != 0;
            double s = 0;
            for (int n = 0; n < lq; ++n) {
                float best = -INFINITY;
                for (int m = 0; m < lp; ++m) {
                    float sim = -1e4f;
                    if (pm[p * lp + m]) {
                        double acc = 0;
                        for (int i = 0; i < d; ++i) acc += (double)from_bf16(Q[(q * lq + n) * d + i]) * from_bf16(P[(p * lp + m) * d + i]);
                        sim = (float)acc;
                    }
                    best = fmaxf(best, sim);
                }
                s += (double)best * (has ? 1.0 : 0.0) * (qm[q * lq + n] ? 1.0 : 0.0);
            }
            want[q * np + p] = (float)s;
        }

    void *dQ, *dP, *dqm, *dpm, *dout, *dws, *dts, *dti;
    CK(hipMalloc(&dQ, Q.size() * 2));
    CK(hipMalloc(&dP, P.size() * 2));
    CK(hipMalloc(&dqm, qm.size()));
    CK(hipMalloc(&dpm, pm.size()));
    CK(hipMalloc(&dout, want.size() * 4));
    CK(hipMalloc(&dts, nq * k * 4));
    CK(hipMalloc(&dti, nq * k * 4));
    const size_t wsb = evdr_maxsim_fwd_workspace(nq, lq, np, lp, EVDR_BF16);
    CK(hipMalloc(&dws, wsb ? wsb : 256));
    CK(hipMemcpy(dQ, Q.data(), Q.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dP, P.data(), P.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dqm, qm.data(), qm.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dpm, pm.data(), pm.size(), hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    if (evdr_version() <= 0) { printf("bad version\n"); return 4; }
    // error path first: wrong width is a status code, not an abort
    if (evdr_maxsim_fwd(dQ, dP, nullptr, nullptr, (float*)dout, nullptr, nq, lq, np, lp, 64, EVDR_BF16, nullptr, dws, wsb, st) != EVDR_ERR_SHAPE) {
        printf("expected EVDR_ERR_SHAPE\n");
        return 5;
    }
    EV(evdr_maxsim_fwd(dQ, dP, (const uint8_t*)dqm, (const uint8_t*)dpm, (float*)dout, nullptr, nq, lq, np, lp, d, EVDR_BF16,
                       nullptr, dws, wsb, st));
    EV(evdr_topk((const float*)dout, nullptr, nq, np, np, 100, k, (float*)dts, (int32_t*)dti, nullptr, 0, st));
    CK(hipStreamSynchronize(st));
    std::vector<float> got(nq * np), ts(nq * k);
    std::vector<int32_t> ti(nq * k);
    CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ts.data(), dts, ts.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ti.data(), dti, ti.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (size_t i = 0; i < got.size(); ++i) worst = fmax(worst, fabs((double)got[i] - want[i]));
    if (worst > 1e-4) { printf("score mismatch %g\n", worst); return 6; }
    for (int q = 0; q < nq; ++q) {
        if (got[q * np + 2] != 0.f) { printf("all-masked page not exactly 0\n"); return 7; }
        for (int j = 0; j < k; ++j) {
            const int p = ti[q * k + j] - 100;
            if (p < 0 || p >= np || ts[q * k + j] != got[q * np + p]) { printf("topk entry wrong\n"); return 8; }
            if (j && ts[q * k + j] > ts[q * k + j - 1]) { printf("topk not sorted\n"); return 9; }
            int better = 0;
            for (int pp = 0; pp < np; ++pp) better += got[q * np + pp] > ts[q * k + j];
            if (better > j) { printf("topk misses a better page\n"); return 10; }
        }
    }
    printf("cabi_smoke OK (max |diff| %.2e)\n", worst);
    return 0;
}
