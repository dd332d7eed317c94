import sys, torch
sys.path.insert(0, "."); import evdr_amd
from evdr_amd import ops
dev = torch.device("cuda:0")
g0 = torch.Generator(device=dev).manual_seed(1)
N, Ls, B, Lq = 500, 206, 32, 32
Q = torch.randn(B, Lq, 128, generator=g0, device=dev)
qm = torch.ones(B, Lq, dtype=torch.bool, device=dev); pm = torch.ones(N, Ls, dtype=torch.bool, device=dev)
arg = torch.randint(0, Ls, (B, N, Lq), generator=g0, device=dev, dtype=torch.int32).to(torch.int16)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
g = torch.randn(B, N, generator=g0, device=dev)
print("full            us", t(lambda: ops.maxsim_backward(g, Q, qm, pm, arg, N, Ls)))
print("g = 0 (no walk) us", t(lambda: ops.maxsim_backward(torch.zeros_like(g), Q, qm, pm, arg, N, Ls)))
print("no pmask        us", t(lambda: ops.maxsim_backward(g, Q, qm, None, arg, N, Ls)))
print("8 queries       us", t(lambda: ops.maxsim_backward(g[:8].contiguous(), Q[:8], qm[:8], pm, arg[:8].contiguous(), N, Ls)))
arg0 = torch.zeros_like(arg)
print("all same row    us", t(lambda: ops.maxsim_backward(g, Q, qm, pm, arg0, N, Ls)))
import time
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(50): ops.maxsim_backward(g, Q, qm, pm, arg, N, Ls)
torch.cuda.synchronize(); print("wall per call us", (time.perf_counter()-t0)/50*1e6)
# skewed argmax distributions: a few patches win most (query, token) pairs of a page
import numpy as np
def zipf():
    return torch.from_numpy(np.random.default_rng(0).zipf(1.5, size=(B, N, Lq)).clip(max=Ls) - 1).long()
def hot(frac):
    return torch.where(torch.rand(B, N, Lq) < frac, torch.zeros(B, N, Lq, dtype=torch.long), torch.randint(0, Ls, (B, N, Lq)))
for name, mk in (("zipf 1.5", zipf), ("30% on one row", lambda: hot(0.3)), ("10% on one row", lambda: hot(0.1))):
    a = mk().to(dev)
    a = ((a + torch.arange(N, device=dev)[None, :, None] * 7) % Ls).to(torch.int16).contiguous()      # the hot row differs per page
    print(f"{name:16s} us", t(lambda: ops.maxsim_backward(g, Q, qm, pm, a, N, Ls)))
