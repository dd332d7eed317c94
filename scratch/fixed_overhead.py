"""Fixed cost of a forward launch at the training shape: time vs pages per workgroup with the workgroup count held at ~250."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
from evdr_amd import _lib as L, ops
dev = torch.device("cuda:0"); torch.manual_seed(0); lib = L.load()
def unit(*s): return torch.nn.functional.normalize(torch.randn(*s, device=dev), dim=-1)
B = 32
Q = unit(B, 32, 128); qp, qa = ops.split_f32(Q)
for name, lp, am in (("teacher", 1030, False), ("student", 206, True)):
    for ppb in (1, 2, 4, 8):
        N = 125 * ppb
        P = unit(N, lp, 128); pp, pa = ops.split_f32(P); tm, pf = ops.pack_pmask(None, N, lp, dev)
        out = torch.empty(B, N, device=dev); arg = torch.empty(B, N, 32, dtype=torch.int16, device=dev) if am else None
        st = L.current_stream_handle(dev)
        lib.evdr_debug_set_pages_per_block(ppb)
        def call():
            L.check(lib.evdr_maxsim_fwd_prepared(L.ptr(qp), L.ptr(pp), None, L.ptr(tm), L.ptr(pf), L.ptr(out), N, L.ptr(arg), B, 32, N, lp, 2, lp * 128, N * lp * 128, L.ptr(qa), L.ptr(pa), None, st))
        for _ in range(10): call()
        torch.cuda.synchronize(); ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30): call()
            b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 30 * 1e3)
        print(f"{name}: {ppb} pages per workgroup, 250 workgroups: {min(ts):7.1f} us", flush=True)
lib.evdr_debug_set_pages_per_block(0)
