"""For given fuzz seeds: where the kernel's argmax differs from the oracle's, the float64 similarity at both indices (a difference at
fp32-noise level = two patches tie; anything larger = a wrong index).  usage: python scratch/fuzz_why.py <seed> [<seed> ...]"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import evdr_amd
from evdr_amd import ops
from oracle import maxsim_oracle as O
import test_gpu_random_sweep as T
dev = "cuda:0"
for seed in map(int, sys.argv[1:]):
    Q, P, qm, pm = T._case(seed)
    if seed % 3 == 0:
        g = torch.Generator().manual_seed(seed)
        nq = int(torch.randint(1, 45, (1,), generator=g)); npg = int(torch.randint(1, 20, (1,), generator=g)); lp = [1030, 1024, 1056, 993, 513][seed % 5]
        Q = torch.nn.functional.normalize(torch.randn(nq, 32, 128, generator=g), dim=-1).bfloat16()
        P = torch.nn.functional.normalize(torch.randn(npg, lp, 128, generator=g), dim=-1).bfloat16()
        a = torch.randint(0, lp, (npg,), generator=g); b = torch.randint(0, lp + 1, (npg,), generator=g)
        lo, hi = torch.minimum(a, b), torch.maximum(a, b)
        ar = torch.arange(lp)[None, :]
        pm = (ar >= lo[:, None]) & (ar < hi[:, None])
        if seed % 2: pm[:, int(torch.randint(0, lp, (1,), generator=g))] = False
        qm = torch.rand(nq, 32, generator=g) > 0.2
    want, warg = O.maxsim_masked_argmax(Q.float(), P.float(), qm, pm)
    s, arg = ops.maxsim_forward(Q.to(dev), P.to(dev), qm.to(dev), pm.to(dev), want_argmax=True)
    arg = (arg.cpu().to(torch.int32) & 0xFFFF)
    diff = (arg != warg.to(torch.int32)).nonzero()
    print(f"seed {seed}: Q{tuple(Q.shape)} P{tuple(P.shape)}: {len(diff)} differing argmax entries of {arg.numel()}")
    for q, p, n in diff[:6].tolist():
        ik, io = int(arg[q, p, n]), int(warg[q, p, n])
        sk = float(Q[q, n].double() @ P[p, ik].double()); so = float(Q[q, n].double() @ P[p, io].double())
        print(f"   (q={q}, page={p}, token={n}): kernel idx {ik} (valid={bool(pm[p, ik])}) sim {sk:.9f} | oracle idx {io} (valid={bool(pm[p, io])}) sim {so:.9f} | diff {sk - so:+.2e}  qmask={bool(qm[q, n])} page has valid={bool(pm[p].any())}")
