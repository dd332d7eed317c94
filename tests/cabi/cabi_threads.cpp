// Several host threads drive the C ABI (include/evdr.h) at once, each on its own stream with its own buffers and its own
// problem shape -- so that different kernel instances see their FIRST launch (the one that raises the instance's dynamic-LDS
// limit and records the device) from racing threads.  Every thread's scores must equal, bit for bit, what the same call gives
// afterwards from one thread alone.  Torch-free: HIP runtime + libevdr.so.  Built and run by tests/test_gpu_cabi.py; prints
// "cabi_threads OK".
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "evdr.h"

struct Case {
    int64_t nq, lq, np, lp;
    int dtype;                      // EVDR_BF16 or EVDR_F32
    void *dQ = nullptr, *dP = nullptr, *dqm = nullptr, *dpm = nullptr, *dout = nullptr, *dws = nullptr;
    size_t wsb = 0;
    hipStream_t st = nullptr;
    std::vector<float> got, alone;
    int rc = 0;
    char err[256] = "";
    int force_variant = 0;          // > 0: this thread forces a forward-kernel family for ITS launches (thread-local debug hook)
    char kname[128] = "";           // instance its last threaded launch dispatched
};

static uint16_t to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static int setup(Case& c, unsigned seed) {
    const int64_t d = 128;
    const size_t esz = c.dtype == EVDR_BF16 ? 2 : 4;
    std::vector<uint8_t> Q(c.nq * c.lq * d * esz), P(c.np * c.lp * d * esz), qm(c.nq * c.lq), pm(c.np * c.lp);
    srand(seed);
    auto rnd = [] { return ((float)rand() / RAND_MAX - 0.5f) * 0.2f; };
    auto fill = [&](std::vector<uint8_t>& v) {
        for (size_t i = 0; i < v.size() / esz; ++i) {
            const float f = rnd();
            if (esz == 2) { const uint16_t b = to_bf16(f); memcpy(&v[i * 2], &b, 2); } else memcpy(&v[i * 4], &f, 4);
        }
    };
    fill(Q);
    fill(P);
    for (auto& v : qm) v = (rand() % 5) != 0;
    for (auto& v : pm) v = (rand() % 6) != 0;
    if (hipMalloc(&c.dQ, Q.size()) || hipMalloc(&c.dP, P.size()) || hipMalloc(&c.dqm, qm.size()) || hipMalloc(&c.dpm, pm.size()) ||
        hipMalloc(&c.dout, c.nq * c.np * 4))
        return 1;
    c.wsb = evdr_maxsim_fwd_workspace(c.nq, c.lq, c.np, c.lp, c.dtype);
    if (hipMalloc(&c.dws, c.wsb ? c.wsb : 256)) return 1;
    if (hipMemcpy(c.dQ, Q.data(), Q.size(), hipMemcpyHostToDevice) || hipMemcpy(c.dP, P.data(), P.size(), hipMemcpyHostToDevice) ||
        hipMemcpy(c.dqm, qm.data(), qm.size(), hipMemcpyHostToDevice) || hipMemcpy(c.dpm, pm.data(), pm.size(), hipMemcpyHostToDevice))
        return 1;
    if (hipStreamCreate(&c.st)) return 1;
    c.got.resize(c.nq * c.np);
    c.alone.resize(c.nq * c.np);
    return 0;
}

static int score(Case& c, std::vector<float>& into) {
    const int rc = evdr_maxsim_fwd(c.dQ, c.dP, (const uint8_t*)c.dqm, (const uint8_t*)c.dpm, (float*)c.dout, nullptr, c.nq, c.lq, c.np,
                                   c.lp, 128, c.dtype, nullptr, c.dws, c.wsb, c.st);
    if (rc != EVDR_OK) {
        snprintf(c.err, sizeof(c.err), "%s", evdr_last_error());       // this thread's own error text
        return rc;
    }
    snprintf(c.kname, sizeof(c.kname), "%s", evdr_last_fwd_kernel());
    if (hipStreamSynchronize(c.st) != hipSuccess) return -1;
    return hipMemcpy(into.data(), c.dout, into.size() * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}

int main() {
    // one problem per kernel family: flat ring (short bf16 pages), staged ring at 1 / 2 / 4 queries per wave, the 4-wave workgroups of
    // 3-12 queries, fp16 hi/lo planes with 3- and 4-tile stages
    std::vector<Case> cases = {
        {5, 9, 13, 75, EVDR_BF16},   {40, 32, 9, 300, EVDR_BF16}, {3, 32, 20, 1030, EVDR_BF16}, {7, 17, 11, 520, EVDR_BF16},
        {12, 32, 10, 206, EVDR_F32}, {6, 32, 7, 1030, EVDR_F32},  {100, 20, 6, 260, EVDR_BF16}, {16, 32, 12, 700, EVDR_BF16},
    };
    for (size_t i = 0; i < cases.size(); ++i)
        if (setup(cases[i], 100 + (unsigned)i)) { printf("setup failed\n"); return 2; }
    // the debug overrides are thread-local: thread 1 forces the flat-ring family for its own launches; every other thread must keep
    // dispatching what the default dispatch picks, and must find its own override still 0 at the end
    cases[1].force_variant = 1;
    std::atomic<int> go{0};
    std::vector<std::thread> th;
    for (auto& c : cases)
        th.emplace_back([&c, &go] {
            while (!go.load()) {}                                        // all threads leave together: first launches race
            if (c.force_variant && evdr_debug_set_fwd_variant(c.force_variant) != 0) c.rc = -5;
            for (int it = 0; it < 25 && c.rc == 0; ++it) c.rc = score(c, c.got);
            if (c.rc == 0 && evdr_debug_set_fwd_variant(0) != c.force_variant) c.rc = -6;      // nobody else touched this thread's override
            // a failing call from this thread must leave ITS text in evdr_last_error (thread-local), whatever the others do
            if (c.rc == 0 && evdr_maxsim_fwd(c.dQ, c.dP, nullptr, nullptr, (float*)c.dout, nullptr, c.nq, c.lq, c.np, c.lp, 64, c.dtype, nullptr,
                                             c.dws, c.wsb, c.st) != EVDR_ERR_SHAPE)
                c.rc = -3;
            if (c.rc == 0 && strstr(evdr_last_error(), "128") == nullptr) c.rc = -4;
        });
    go.store(1);
    for (auto& t : th) t.join();
    for (size_t i = 0; i < cases.size(); ++i)
        if (cases[i].rc != 0) { printf("thread %zu failed: rc %d %s\n", i, cases[i].rc, cases[i].err); return 3; }
    if (evdr_debug_set_fwd_variant(0) != 0) { printf("a worker thread's override leaked into the main thread\n"); return 7; }
    for (size_t i = 0; i < cases.size(); ++i) {
        Case& c = cases[i];
        char threaded[128];
        snprintf(threaded, sizeof(threaded), "%s", c.kname);
        if (score(c, c.alone) != 0) { printf("serial call %zu failed: %s\n", i, c.err); return 4; }
        if (c.force_variant) {
            // forced family in the thread, default family here: different instances, the same scores to rounding
            if (strncmp(threaded, "maxsim_fwd16_kernel<", 20) != 0 || strcmp(threaded, c.kname) == 0) {
                printf("case %zu: forced variant dispatched %s (serial: %s)\n", i, threaded, c.kname);
                return 8;
            }
            for (size_t e = 0; e < c.got.size(); ++e)
                if (!(c.got[e] - c.alone[e] < 1e-5f && c.alone[e] - c.got[e] < 1e-5f)) { printf("case %zu: forced variant scores differ\n", i); return 5; }
            continue;
        }
        if (strcmp(threaded, c.kname) != 0) { printf("case %zu: dispatched %s next to a thread that forced a variant, %s alone\n", i, threaded, c.kname); return 9; }
        if (memcmp(c.got.data(), c.alone.data(), c.got.size() * 4) != 0) { printf("case %zu: threaded scores differ from the serial ones\n", i); return 5; }
        bool nonzero = false;
        for (float v : c.got) nonzero |= v != 0.f;
        if (!nonzero) { printf("case %zu: all scores zero\n", i); return 6; }
    }
    printf("cabi_threads OK (%zu threads)\n", cases.size());
    return 0;
}
