"""Backward + AdamW with and without the next step's planes, against the separate normalise pass it replaces (training shape)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
from evdr_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
def unit(*s): return torch.nn.functional.normalize(torch.randn(*s, device=dev), dim=-1)
B, N, Ls = 32, 500, 206
Q, x = unit(B, 32, 128), unit(N, Ls, 128)
ea, es = torch.zeros_like(x), torch.zeros_like(x)
g = torch.randn(B, N, device=dev) * 1e-2
arg = torch.randint(0, Ls, (B, N, 32), device=dev).to(torch.int16)
pm = torch.ones(N, Ls, dtype=torch.bool, device=dev)
tm, pf = ops.pack_pmask(pm, N, Ls, dev)
planes = ops.l2norm_split(x, pm, 1e-12, pageflags=pf)
def t(fn, n=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
plain = lambda: ops.maxsim_backward_adamw(g, Q, None, pm, arg, x, ea, es, 1e-3, (0.9, 0.999), 1e-8, 1e-2, 1)
withp = lambda: ops.maxsim_backward_adamw(g, Q, None, pm, arg, x, ea, es, 1e-3, (0.9, 0.999), 1e-8, 1e-2, 1, next_planes=planes, pageflags=pf)
norm = lambda: ops.l2norm_split(x, pm, 1e-12, pageflags=pf, out=planes)
for rnd in range(3):
    print(f"bwd+adamw {t(plain):6.1f} us   bwd+adamw+planes {t(withp):6.1f} us   l2norm_split {t(norm):6.1f} us   both {t(lambda: (plain(), norm())):6.1f} us", flush=True)
