"""One-off fuzz of the forward kernels against the oracle: many seeds of tests/test_gpu_random_sweep.py's case generator plus
1030-patch range/hole layouts with random query counts; every fifth seed at another embedding width (round 6: 129..256 columns run on
two column blocks, narrower ones on zero columns).  usage: python scratch/fuzz_fwd.py <first_seed> <count> [long]
`long` (round 3): every case is a long-page case -- lp drawn from 1057 ... 65535 (random lengths, powers of two and their neighbours,
the ABI bound), range / hole layouts whose range may start beyond patch 4095, few pages."""
import os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import evdr_amd
from evdr_amd import _lib as _L
if os.environ.get("EVDR_FUZZ_LIB"): _L.LIB_PATH = os.path.join(_L.PKG_DIR, os.environ["EVDR_FUZZ_LIB"])     # e.g. libevdr_sentinel.so (scratch-only switch)
from evdr_amd import ops
if os.environ.get("EVDR_FUZZ_VARIANT"): _L.load().evdr_debug_set_fwd_variant(int(os.environ["EVDR_FUZZ_VARIANT"]))   # e.g. 33: the nt stream of the fp16-plane forward forced on (scratch-only switch)
from oracle import maxsim_oracle as O
import test_gpu_random_sweep as T
dev = "cuda:0"; s0, n = int(sys.argv[1]), int(sys.argv[2]); bad = 0
LONG = len(sys.argv) > 3 and sys.argv[3] == "long"
torch.set_num_threads(16)
for seed in range(s0, s0 + n):
    Q, P, qm, pm = T._case(seed)
    if seed % 3 == 0 or LONG:                           # long pages with range / hole layouts and random query counts
        g = torch.Generator().manual_seed(seed)
        nq = int(torch.randint(1, 45, (1,), generator=g)); npg = int(torch.randint(1, 20, (1,), generator=g)); lp = [1030, 1024, 1056, 993, 513][seed % 5]
        if LONG:
            lp = [int(torch.randint(1057, 9000, (1,), generator=g)), 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8193, 16385, 32767, 32769,
                  int(torch.randint(9000, 65536, (1,), generator=g)), 65535][seed % 14]
            npg = int(torch.randint(1, 6 if lp > 9000 else 12, (1,), generator=g))
            nq = int(torch.randint(1, 20 if lp > 9000 else 45, (1,), generator=g))
        Q = torch.nn.functional.normalize(torch.randn(nq, 32, 128, generator=g), dim=-1).bfloat16()
        P = torch.nn.functional.normalize(torch.randn(npg, lp, 128, generator=g), dim=-1).bfloat16()
        a = torch.randint(0, lp, (npg,), generator=g); b = torch.randint(0, lp + 1, (npg,), generator=g)
        lo, hi = torch.minimum(a, b), torch.maximum(a, b)
        ar = torch.arange(lp)[None, :]
        pm = (ar >= lo[:, None]) & (ar < hi[:, None])
        if seed % 2: pm[:, int(torch.randint(0, lp, (1,), generator=g))] = False      # one hole: mask words decide
        qm = torch.rand(nq, 32, generator=g) > 0.2
    if seed % 5 == 4 and not LONG:                      # another embedding width: the same case re-drawn at d columns (fp32 and bf16 inputs)
        g = torch.Generator().manual_seed(seed + 1)
        d = [256, 200, 129, 255, 64, 256][seed % 6]
        Q = torch.nn.functional.normalize(torch.randn(Q.shape[0], Q.shape[1], d, generator=g), dim=-1).bfloat16()
        P = torch.nn.functional.normalize(torch.randn(P.shape[0], P.shape[1], d, generator=g), dim=-1).bfloat16()
    want, warg = O.maxsim_masked_argmax(Q.float(), P.float(), qm, pm)
    for am in (False, True):
        for Qx, Px in ((Q, P), (Q.float(), P.float())):
            s, arg = ops.maxsim_forward(Qx.to(dev), Px.to(dev), qm.to(dev), pm.to(dev), want_argmax=am)
            err = (s.cpu() - want).abs().max().item()
            ok = err < 1e-4
            if ok and am:
                got = arg.cpu().to(torch.int64) & 0xFFFF
                for q, p_, t in (got != warg.to(torch.int64)).nonzero().tolist():
                    # two patches within fp32 noise of each other are a tie (either index is a maximiser); anything else is wrong
                    sk = float(Q[q, t].double() @ P[p_, int(got[q, p_, t])].double())
                    so = float(Q[q, t].double() @ P[p_, int(warg[q, p_, t])].double())
                    if abs(sk - so) > 2e-7 or not bool(pm[p_, int(got[q, p_, t])]):
                        ok = False
            if not ok:
                bad += 1; print(f"FAIL seed={seed} argmax={am} dtype={Qx.dtype} shape Q{tuple(Q.shape)} P{tuple(P.shape)} err={err:.3e}", flush=True)
    if (seed - s0) % 50 == 49: print(f"... {seed - s0 + 1} seeds, {bad} failures", flush=True)
print(f"done: {n} seeds, {bad} failures")
