"""Student-forward shape (fp32, argmax, 206 patches): 3-tile vs 4-tile stages, interleaved repeats to cancel clock drift."""
import _hooks as H
import os, sys, torch
sys.path.insert(0, "."); import evdr_amd
from evdr_amd import ops, _lib as L
dev = torch.device("cuda:0"); torch.manual_seed(0); lib = L.load()
def unit(*s): return torch.nn.functional.normalize(torch.randn(*s, device=dev), dim=-1)
for nq, np_, lp, am in [(32, 500, 206, True), (32, 500, 206, False), (500, 6847, 206, False), (32, 500, 700, False), (32, 500, 1030, False)]:
    Q, P = unit(nq, 32, 128), unit(np_, lp, 128)
    qp, qa = ops.split_f32(Q); pp, pa = ops.split_f32(P)
    tm, pf = ops.pack_pmask(None, np_, lp, dev)
    out = torch.empty(nq, np_, device=dev); arg = torch.empty(nq, np_, 32, dtype=torch.int16, device=dev) if am else None
    st = L.current_stream_handle(dev)
    def call():
        L.check(lib.evdr_maxsim_fwd_prepared(L.ptr(qp), L.ptr(pp), None, L.ptr(tm), L.ptr(pf), L.ptr(out), np_, L.ptr(arg), nq, 32, np_, lp, 2,
                                             lp * 128, np_ * lp * 128, L.ptr(qa), L.ptr(pa), None, st))
    tot = {"10": 0.0, "11": 0.0}; ref = None
    for rep in range(6):
        for v in ("10", "11"):
            H.set_variant(v)
            for _ in range(20): call()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50 if nq < 100 else 3
            a.record()
            for _ in range(n): call()
            b.record(); torch.cuda.synchronize(); tot[v] += a.elapsed_time(b) / n
            if ref is None: ref = out.clone()
            assert torch.equal(ref, out)
    print(f"nq={nq:4d} np={np_:5d} lp={lp:5d} argmax={int(am)}  ST=3: {tot['10']/6*1e3:9.1f} us   ST=4: {tot['11']/6*1e3:9.1f} us", flush=True)
