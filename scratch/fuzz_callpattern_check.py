"""For given seeds of scratch/fuzz_callpattern.py: errors of the GPU path AND of the fp32 oracle against a float64 evaluation of the same
formulas (is a gate violation an error of the kernels, or the fp32 noise both sides share?)."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import evdr_amd  # noqa
from evdr_amd.criterion import infonce_distillation_loss
from evdr_amd.evaluator.retrieval import score_multi_vector_masked
from evdr_amd.utils.preprocess_data import l2_normalize
from oracle import maxsim_oracle as O
import test_gpu_random_sweep as T
dev = "cuda:0"
for seed in [int(a) for a in sys.argv[1:]]:
    Q, P, qm, pm = T._case(seed)
    g = torch.Generator().manual_seed(seed)
    Q = Q.float(); X = torch.randn(P.shape, generator=g) * (0.2 + 3.0 * torch.rand(P.shape[0], 1, 1, generator=g))
    if not pm.any(dim=1).all(): pm[~pm.any(dim=1), 0] = True
    Tt = torch.randn(Q.shape[0], P.shape[0], generator=g)
    def oracle(dt):
        Xo = X.clone().to(dt).requires_grad_(True)
        so = O.maxsim_masked(Q.to(dt), O.l2_normalize(Xo * pm.unsqueeze(-1)), qm, pm)
        lo = O.infonce_distill(so, Tt.to(dt), 0.1); lo.backward()
        return so.detach(), lo.detach(), Xo.grad
    s64, l64, g64 = oracle(torch.float64); s32, l32, g32 = oracle(torch.float32)
    Xd = X.clone().to(dev).requires_grad_(True)
    s = score_multi_vector_masked(Q.to(dev), l2_normalize(Xd * pm.to(dev).unsqueeze(-1)), qm.to(dev), pm.to(dev))
    loss = infonce_distillation_loss(s, Tt.to(dev), temperature=0.1); loss.backward()
    print(f"seed {seed}: scores |gpu-f64| {(s.detach().cpu().double() - s64).abs().max():.2e}  |f32-f64| {(s32.double() - s64).abs().max():.2e};  "
          f"loss rel gpu {abs(loss.item() - l64.item()) / abs(l64.item()):.2e}  f32 {abs(l32.item() - l64.item()) / abs(l64.item()):.2e};  "
          f"grad |gpu-f64| {(Xd.grad.cpu().double() - g64).abs().max():.2e}  |f32-f64| {(g32.double() - g64).abs().max():.2e}  max|grad| {g64.abs().max():.2e}", flush=True)
