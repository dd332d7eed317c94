"""Student / teacher forward shapes (fp32 as fp16 planes, 32 queries x 500 pages): the shipped two queries per wave (variant 0) against ONE
query per wave on the same 8-wave workgroups (variant 35, experiment build: 8 queries per workgroup, four query groups, half the query
bytes per workgroup, twice the LDS fragment reads per FLOP).  Interleaved repeats; bit-identical results required."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
from evdr_amd import _lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.PKG_DIR), "scratch", "_variants", "libevdr_exp.so")
from evdr_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0); lib = L.load()
def unit(*s): return torch.nn.functional.normalize(torch.randn(*s, device=dev), dim=-1)
for nq, np_, lp, am in [(32, 500, 206, True), (32, 500, 1030, False), (32, 125, 206, True), (32, 63, 206, True), (32, 63, 1030, False), (32, 32, 1030, False), (32, 100, 206, True), (16, 63, 1030, False)]:
    Q, P = unit(nq, 32, 128), unit(np_, lp, 128)
    qp, qa = ops.split_f32(Q); pp, pa = ops.split_f32(P)
    tm, pf = ops.pack_pmask(None, np_, lp, dev)
    out = torch.empty(nq, np_, device=dev); arg = torch.empty(nq, np_, 32, dtype=torch.int16, device=dev) if am else None
    st = L.current_stream_handle(dev)
    def call():
        L.check(lib.evdr_maxsim_fwd_prepared(L.ptr(qp), L.ptr(pp), None, L.ptr(tm), L.ptr(pf), L.ptr(out), np_, L.ptr(arg), nq, 32, np_, lp, 2,
                                             lp * 128, np_ * lp * 128, L.ptr(qa), L.ptr(pa), None, st))
    variants = (0, 35)
    tot = {v: 0.0 for v in variants}; ref = None; names = {}
    for rep in range(6):
        for v in variants:
            lib.evdr_debug_set_fwd_variant(v)
            for _ in range(20): call()
            torch.cuda.synchronize()
            names[v] = lib.evdr_last_fwd_kernel().decode()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50): call()
            b.record(); torch.cuda.synchronize(); tot[v] += a.elapsed_time(b) / 50
            if ref is None: ref = (out.clone(), arg.clone() if am else None)
            assert torch.equal(ref[0], out) and (not am or torch.equal(ref[1], arg)), f"variant {v} differs"
    lib.evdr_debug_set_fwd_variant(0)
    for v in variants:
        print(f"nq={nq:3d} np={np_:4d} lp={lp:4d} argmax={int(am)}  variant {v:2d} {names[v]:62s} {tot[v] / 6 * 1e3:8.1f} us", flush=True)
