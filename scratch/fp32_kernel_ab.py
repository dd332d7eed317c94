"""Kernel-only time of the fp16 hi/lo forward on prepared planes (no split, no mask packing), per EVDR_FWD_VARIANT."""
import _hooks as H
import os, sys, torch
sys.path.insert(0, "."); import evdr_amd
from evdr_amd import ops, _lib as L
dev = torch.device("cuda:0")
torch.manual_seed(0)
lib = L.load()
def unit(*s): return torch.nn.functional.normalize(torch.randn(*s, device=dev), dim=-1)
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"])]
for nq, np_, lp, am in [(32, 500, 1030, False), (32, 500, 206, True), (32, 500, 206, False), (500, 6847, 1030, False), (500, 500, 1030, False), (32, 500, 1030, True), (64, 500, 206, True)]:
    Q, P = unit(nq, 32, 128), unit(np_, lp, 128)
    qp, qa = ops.split_f32(Q); pp, pa = ops.split_f32(P)
    tm, pf = ops.pack_pmask(None, np_, lp, dev)
    out = torch.empty(nq, np_, device=dev); arg = torch.empty(nq, np_, 32, dtype=torch.int16, device=dev) if am else None
    st = L.current_stream_handle(dev)
    def call():
        L.check(lib.evdr_maxsim_fwd_prepared(L.ptr(qp), L.ptr(pp), None, L.ptr(tm), L.ptr(pf), L.ptr(out), np_, L.ptr(arg), nq, 32, np_, lp, 2,
                                             lp * 128, np_ * lp * 128, L.ptr(qa), L.ptr(pa), None, st))
    line = f"nq={nq:4d} np={np_:5d} lp={lp:5d} argmax={int(am)}"
    for v in variants:
        H.set_variant(v)
        call(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        a.record()
        for _ in range(reps): call()
        b.record(); torch.cuda.synchronize(); ms = a.elapsed_time(b) / reps
        line += f" | v{v}: {ms*1e3:9.1f} us {nq*np_*2*32*lp*128*3/ms/1e9:7.1f} TF"
    print(line, flush=True)
