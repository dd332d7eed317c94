"""Student / teacher forward at the training shape vs pages per workgroup (debug hook) and queries per wave."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
from evdr_amd import _lib as L, ops
dev = torch.device("cuda:0"); torch.manual_seed(0); lib = L.load()
def unit(*s): return torch.nn.functional.normalize(torch.randn(*s, device=dev), dim=-1)
B, N = 32, 500
Q = unit(B, 32, 128); qp, qa = ops.split_f32(Q)
for name, lp, am in (("student", 206, True), ("teacher", 1030, False)):
    P = unit(N, lp, 128); pp, pa = ops.split_f32(P); tm, pf = ops.pack_pmask(None, N, lp, dev)
    out = torch.empty(B, N, device=dev); arg = torch.empty(B, N, 32, dtype=torch.int16, device=dev) if am else None
    st = L.current_stream_handle(dev)
    def call():
        L.check(lib.evdr_maxsim_fwd_prepared(L.ptr(qp), L.ptr(pp), None, L.ptr(tm), L.ptr(pf), L.ptr(out), N, L.ptr(arg), B, 32, N, lp, 2, lp * 128, N * lp * 128, L.ptr(qa), L.ptr(pa), None, st))
    for ppb in (0, 1, 2, 3, 4, 6, 8):
        lib.evdr_debug_set_pages_per_block(ppb)
        for _ in range(10): call()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30): call()
            b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 30 * 1e3)
        print(f"{name} ppb={ppb}: {min(ts):7.1f} us  ({lib.evdr_last_fwd_kernel().decode()})", flush=True)
    lib.evdr_debug_set_pages_per_block(0)
