"""In-kernel stamp shares of the teacher forward (fp16 planes, no argmax, 1030 patches) at the training shape (experiment build)."""
import os, sys, torch
import _hooks as H
EXP = H.use_experiment_build()
import evdr_amd
from evdr_amd import _lib as L, ops
dev = torch.device("cuda:0"); torch.manual_seed(0); lib = L.load()
def unit(*s): return torch.nn.functional.normalize(torch.randn(*s, device=dev), dim=-1)
B, N, lp = 32, 500, 1030
Q, P = unit(B, 32, 128), unit(N, lp, 128)
qp, qa = ops.split_f32(Q); pp, pa = ops.split_f32(P); tm, pf = ops.pack_pmask(None, N, lp, dev)
out = torch.empty(B, N, device=dev); dbg = torch.zeros((4096, 8, 8), dtype=torch.int64, device=dev)
EXP.evdr_experiment_set_dbg_buffer(dbg.data_ptr()); H.set_variant(53)
st = L.current_stream_handle(dev)
for _ in range(3):
    L.check(lib.evdr_maxsim_fwd_prepared(L.ptr(qp), L.ptr(pp), None, L.ptr(tm), L.ptr(pf), L.ptr(out), N, None, B, 32, N, lp, 2, lp * 128, N * lp * 128, L.ptr(qa), L.ptr(pa), None, st))
torch.cuda.synchronize(); print(lib.evdr_last_fwd_kernel().decode())
d = dbg.cpu().double(); d = d[d[:, :, 0] > 0]; tot = d[:, 0]
names = ["total", "prologue", "barrier wait", "refill at stage top", "fast block", "generic / tail tile", "page finish", "control before barrier"]
print(f"waves reporting: {len(d)}, mean total cycles {tot.mean():.0f} (4 (page, query group) units per workgroup, 8 stages + tail per page)")
for i, n in enumerate(names[1:], 1):
    print(f"{n:22s} {100*(d[:, i]/tot).mean():6.2f} %   ({d[:, i].mean():9.0f} cycles per workgroup)")
print(f"{'unaccounted':22s} {100*(1 - (d[:,1:8].sum(1)/tot)).mean():6.2f} %")
