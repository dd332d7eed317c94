"""Fuzz of the reference's CALL PATTERN through the drop-in modules with the planes hand-over: Psb = l2_normalize(X * pmask[..., None]) ->
score_multi_vector_masked(Q, Psb, qmask, pmask) -> infonce_distillation_loss against teacher scores -> backward to X, against the oracle's
autograd evaluated in FLOAT64 (scores 1e-4; loss 1e-5 relative and dX 2e-6 + 1e-5 relative, or three times the fp32 oracle's own distance from
the float64 result where that is larger: the l2-normalise backward cancels two terms, and a single-row loss has no averaging); random shapes / masks of tests/test_gpu_random_sweep.py's generator
(fp32).  Every seed also checks that the scorer really took the planes l2_normalize left (ops.planes_of) and that a second scoring of the
same tensors gives the same bits.  usage: python scratch/fuzz_callpattern.py <first_seed> <count>"""
import os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import evdr_amd  # noqa
from evdr_amd import ops
from evdr_amd.criterion import infonce_distillation_loss
from evdr_amd.evaluator.retrieval import score_multi_vector_masked
from evdr_amd.utils.preprocess_data import l2_normalize
from oracle import maxsim_oracle as O
import test_gpu_random_sweep as T
dev = "cuda:0"; s0, n = int(sys.argv[1]), int(sys.argv[2]); bad = 0
torch.set_num_threads(16)
for seed in range(s0, s0 + n):
    Q, P, qm, pm = T._case(seed)
    g = torch.Generator().manual_seed(seed)
    Q = Q.float(); X = torch.randn(P.shape, generator=g) * (0.2 + 3.0 * torch.rand(P.shape[0], 1, 1, generator=g))
    if not pm.any(dim=1).all():
        pm[~pm.any(dim=1), 0] = True                              # (an all-masked page has no gradient to compare)
    Tt = torch.randn(Q.shape[0], P.shape[0], generator=g)
    def oracle(dt):
        Xo = X.clone().to(dt).requires_grad_(True)
        so = O.maxsim_masked(Q.to(dt), O.l2_normalize(Xo * pm.unsqueeze(-1)), qm, pm)
        lo = O.infonce_distill(so, Tt.to(dt), 0.1); lo.backward()
        return so.detach(), lo.detach(), Xo.grad
    so, lo, go = oracle(torch.float64)                            # the yardstick
    s32, l32, g32 = oracle(torch.float32)                         # the fp32 oracle's own distance from it: the noise both fp32 paths share
    Xd = X.clone().to(dev).requires_grad_(True)
    Psb = l2_normalize(Xd * pm.to(dev).unsqueeze(-1))
    took = ops.planes_of(Psb) is not None
    s = score_multi_vector_masked(Q.to(dev), Psb, qm.to(dev), pm.to(dev))
    s2 = score_multi_vector_masked(Q.to(dev), Psb, qm.to(dev), pm.to(dev))
    loss = infonce_distillation_loss(s, Tt.to(dev), temperature=0.1); loss.backward()
    es = (s.detach().cpu().double() - so).abs().max().item()
    el = abs(loss.item() - lo.item()) / max(1.0, abs(lo.item())); el32 = abs(l32.item() - lo.item()) / max(1.0, abs(lo.item()))
    eg = ((Xd.grad.cpu().double() - go).abs() - 1e-5 * go.abs()).max().item(); eg32 = ((g32.double() - go).abs() - 1e-5 * go.abs()).max().item()
    ok = took and es < 1e-4 and el < max(1e-5, 3 * el32) and eg < max(2e-6, 3 * eg32) and torch.equal(s.detach(), s2.detach())
    if not ok:
        bad += 1; print(f"seed {seed}: took={took} |ds|={es:.2e} dloss={el:.2e} (fp32 oracle {el32:.2e}) dgrad-excess={eg:.2e} (fp32 oracle {eg32:.2e}) "
                        f"shapes Q{tuple(Q.shape)} P{tuple(P.shape)}", flush=True)
    if (seed - s0 + 1) % 50 == 0: print(f"... {seed - s0 + 1} seeds, {bad} failures", flush=True)
print(f"done: {n} seeds, {bad} failures")
