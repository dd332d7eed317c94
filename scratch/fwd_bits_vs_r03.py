"""Forward results of this round's library against the round-3 library (scratch/ab/libevdr_r03.so, built from commit 895a2d0), bit for
bit: scores and arg-max of evdr_maxsim_fwd over random shapes, masks, dtypes and query counts (every queries-per-wave regime, both ring
kinds).  The ring hand-over and the page-flag prefetch changed; no arithmetic did -- so every bit must agree.
usage: python scratch/fwd_bits_vs_r03.py <first_seed> <count>"""
import ctypes as C, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import evdr_amd
from evdr_amd import _lib as L
import test_gpu_random_sweep as T
dev = torch.device("cuda:0"); s0, n = int(sys.argv[1]), int(sys.argv[2])
libs = []
for p in (os.path.join(R, "scratch", "ab", "libevdr_r03.so"), L.LIB_PATH):
    lib = C.CDLL(p)
    for name in ("evdr_maxsim_fwd", "evdr_maxsim_fwd_workspace"):
        getattr(lib, name).restype, getattr(lib, name).argtypes = L.SIGNATURES[name]
    libs.append(lib)
stream = torch.cuda.current_stream(dev).cuda_stream
bad = 0
for seed in range(s0, s0 + n):
    Q, P, qm, pm = T._case(seed)
    g = torch.Generator().manual_seed(seed)
    if seed % 2:                                         # 1030-patch pages, 1 .. 70 queries: the staged kernel's regimes
        nq, npg = int(torch.randint(1, 71, (1,), generator=g)), int(torch.randint(1, 40, (1,), generator=g))
        Q = torch.nn.functional.normalize(torch.randn(nq, 32, 128, generator=g), dim=-1).bfloat16()
        P = torch.nn.functional.normalize(torch.randn(npg, 1030, 128, generator=g), dim=-1).bfloat16()
        lens = torch.randint(600, 1031, (npg,), generator=g)
        pm = torch.arange(1030)[None, :] < lens[:, None]
        if seed % 4 == 1: pm[:, :5] = False
        qm = torch.rand(nq, 32, generator=g) > 0.2
    for dtype, Qx, Px in ((L.EVDR_BF16, Q.bfloat16(), P.bfloat16()), (L.EVDR_F32, Q.float(), P.float())):
        for want_arg in (False, True):
            Qd, Pd, qmd, pmd = Qx.contiguous().to(dev), Px.contiguous().to(dev), qm.to(dev).contiguous(), pm.to(dev).contiguous()
            nq, lq, _ = Qd.shape; npg, lp, _ = Pd.shape
            res = []
            for lib in libs:
                out = torch.full((nq, npg), float("nan"), device=dev)
                arg = torch.full((nq, npg, lq), -1, dtype=torch.int16, device=dev) if want_arg else None
                ws = torch.empty(max(lib.evdr_maxsim_fwd_workspace(nq, lq, npg, lp, dtype), 256), dtype=torch.uint8, device=dev)
                rc = lib.evdr_maxsim_fwd(Qd.data_ptr(), Pd.data_ptr(), qmd.data_ptr(), pmd.data_ptr(), out.data_ptr(),
                                         arg.data_ptr() if want_arg else None, nq, lq, npg, lp, 128, dtype, None, ws.data_ptr(), ws.numel(), stream)
                assert rc == 0, rc
                torch.cuda.synchronize()
                res.append((out, arg))
            same = torch.equal(res[0][0].view(torch.int32), res[1][0].view(torch.int32)) and (not want_arg or torch.equal(res[0][1], res[1][1]))
            if not same:
                bad += 1; print(f"DIFF seed={seed} dtype={dtype} argmax={want_arg} nq={nq} lq={lq} np={npg} lp={lp}", flush=True)
    if (seed - s0) % 50 == 49: print(f"... {seed - s0 + 1} seeds, {bad} differing cases", flush=True)
print(f"done: {n} seeds x 4 (dtype, argmax) cases, {bad} differing from the round-3 library")
