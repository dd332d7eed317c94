"""dQ kernel (evdr_maxsim_bwd_q) time and effective gather rate at the training shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
from evdr_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
for N, L in ((500, 206), (500, 1030)):
    B, Lq = 32, 32
    P = torch.randn(N, L, 128, device=dev)
    g = torch.randn(B, N, device=dev)
    arg = torch.randint(0, L, (B, N, Lq), device=dev, dtype=torch.int32).to(torch.int16)
    qm = torch.ones(B, Lq, dtype=torch.bool, device=dev); pm = torch.ones(N, L, dtype=torch.bool, device=dev)
    ops.maxsim_backward_q(g, P, qm, pm, arg, B, Lq); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): ops.maxsim_backward_q(g, P, qm, pm, arg, B, Lq)
    b.record(); torch.cuda.synchronize(); us = a.elapsed_time(b) / 20 * 1e3
    print(f"N={N} L={L}: {us:8.1f} us   gathered {B*Lq*N*512/1e6:.0f} MB -> {B*Lq*N*512/us/1e6:.2f} TB/s", flush=True)
