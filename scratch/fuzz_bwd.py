"""Fuzz of the backward gather (evdr_maxsim_bwd) and of the fused update (evdr_maxsim_bwd_adamw_planes) against plain torch on the
GPU: random page counts / lengths (both sides of the 128-row slab and of the 1024-pair chunk), masks, hot rows.
usage: python scratch/fuzz_bwd.py <first_seed> <count> [long]     (long, round 3: page lengths 1030 ... 65535, few pages)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
from evdr_amd import _lib as _L
if os.environ.get("EVDR_FUZZ_LIB"): _L.LIB_PATH = os.path.join(_L.PKG_DIR, os.environ["EVDR_FUZZ_LIB"])     # e.g. libevdr_sentinel.so (scratch-only switch)
from evdr_amd import ops
dev = torch.device("cuda:0"); s0, n = int(sys.argv[1]), int(sys.argv[2]); bad = 0
LONG = len(sys.argv) > 3 and sys.argv[3] == "long"
for seed in range(s0, s0 + n):
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    npg, lp, nq, lq = ri(1, 9), [1, 7, 40, 127, 128, 129, 206, 300, 513][ri(0, 8)], ri(1, 70), [1, 5, 32, 33][ri(0, 3)]
    if LONG:
        lp = [1030, 2049, 4100, 8200, 20000, 32769, 65535][seed % 7]
        npg = ri(1, 3 if lp > 9000 else 6)
    Q = torch.randn(nq, lq, 128, generator=g).to(dev)
    gr = (torch.randn(nq, npg, generator=g) * (0.0 if seed % 11 == 0 else 1.0)).to(dev)
    pm = (torch.rand(npg, lp, generator=g) > [0.0, 0.3, 0.9][ri(0, 2)]); qm = torch.rand(nq, lq, generator=g) > 0.25
    if npg > 1: pm[ri(0, npg - 1)] = False
    arg = torch.randint(0, lp, (nq, npg, lq), generator=g)
    if seed % 3 == 0: arg[:, :, : max(1, lq // 2)] = ri(0, lp - 1)                 # a hot row
    pm, qm, arg16 = pm.to(dev), qm.to(dev), arg.to(torch.int16).to(dev)
    has = pm.any(dim=1).float()
    w = gr[:, :, None] * qm[:, None, :].float() * has[None, :, None]                       # (nq, npg, lq)
    want = torch.zeros(npg, lp, 128, device=dev, dtype=torch.float64)
    idx = (torch.arange(npg, device=dev)[None, :, None] * lp + arg.to(dev)).reshape(-1)
    want.view(-1, 128).index_add_(0, idx, (w[..., None].double() * Q[:, None, :, :].double()).reshape(-1, 128))
    got = ops.maxsim_backward(gr, Q, qm, pm, arg16, npg, lp)
    e1 = (got.double() - want).abs().max().item()
    again = ops.maxsim_backward(gr, Q, qm, pm, arg16, npg, lp)                              # round 4: bit-reproducible
    if not torch.equal(got, again): e1 = float("inf")
    tol = 1e-5 * max(1.0, want.abs().max().item())
    # fused update with planes == unfused pieces: l2norm backward of `want` + torch AdamW formula, and planes == l2norm_split(x_new)
    x = (torch.randn(npg, lp, 128, generator=g) * 0.5).to(dev) * pm.unsqueeze(-1)
    ea, es = torch.zeros_like(x), torch.zeros_like(x)
    tm, pf = ops.pack_pmask(pm, npg, lp, dev)
    planes = ops.l2norm_split(x, pm, 1e-12, pageflags=pf)
    x_ref = x.clone().requires_grad_(True)
    y = ops.l2norm_forward(x_ref.detach(), pm, 1e-12)[0]
    nrm = (x_ref * pm.unsqueeze(-1)).norm(dim=-1, keepdim=True)
    yy = (x_ref * pm.unsqueeze(-1)) / (nrm + 1e-12)
    yy.backward(want.float())
    opt_x = x.clone().requires_grad_(True); opt = torch.optim.AdamW([opt_x], lr=1e-3, weight_decay=1e-2)
    opt_x.grad = x_ref.grad * pm.unsqueeze(-1); opt.step()
    ops.maxsim_backward_adamw(gr, Q, qm, pm, arg16, x, ea, es, 1e-3, (0.9, 0.999), 1e-8, 1e-2, 1, next_planes=planes, pageflags=pf)
    d = (x - opt_x.detach()).abs()
    big = opt_x.grad.abs() > 1e-6 * max(1.0, float(opt_x.grad.abs().max()))
    e2 = d[big].max().item() if big.any() else 0.0
    fresh = ops.l2norm_split(x, pm, 1e-12)
    ok = e1 <= tol and e2 <= 3e-6 and d.max().item() < 2.1e-3 and torch.equal(planes[0].view(torch.int16), fresh[0].view(torch.int16))
    if not ok:
        bad += 1; print(f"FAIL seed={seed} np={npg} lp={lp} nq={nq} lq={lq}: dP err {e1:.2e} (tol {tol:.1e}), param err {e2:.2e} / {d.max().item():.2e}", flush=True)
    if (seed - s0) % 50 == 49: print(f"... {seed - s0 + 1} seeds, {bad} failures", flush=True)
print(f"done: {n} seeds, {bad} failures")
