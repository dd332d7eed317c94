"""Where the student forward's time per page goes, by differences between launches of the SAME kernel family (fp32 as fp16 planes, 32 queries,
product library): page lengths 96 (one 3-tile stage), 192 (two stages, no tail), 206 (two stages + 14-patch tail tile), 224 (7 full
tiles), 384 (four stages); with and without arg-max; 125 pages (ONE page per workgroup: fixed cost + one page) and 500 (four pages)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
from evdr_amd import _lib as L
if os.environ.get("EVDR_AB_LIB"): L.LIB_PATH = os.path.abspath(os.environ["EVDR_AB_LIB"])     # a variant build (scratch/build_variant.sh)
from evdr_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0); lib = L.load()
def unit(*s): return torch.nn.functional.normalize(torch.randn(*s, device=dev), dim=-1)
Q = unit(32, 32, 128); qp, qa = ops.split_f32(Q)
res = {}
cases = [(np_, lp, am) for lp in (96, 192, 206, 224, 384) for am in (False, True) for np_ in (125, 500)]
bufs = {}
for np_, lp, am in cases:
    P = unit(np_, lp, 128); pp, pa = ops.split_f32(P); tm, pf = ops.pack_pmask(None, np_, lp, dev)
    out = torch.empty(32, np_, device=dev); arg = torch.empty(32, np_, 32, dtype=torch.int16, device=dev) if am else None
    bufs[(np_, lp, am)] = (pp, pa, tm, pf, out, arg)
st = L.current_stream_handle(dev)
def call(k):
    np_, lp, am = k; pp, pa, tm, pf, out, arg = bufs[k]
    L.check(lib.evdr_maxsim_fwd_prepared(L.ptr(qp), L.ptr(pp), None, L.ptr(tm), L.ptr(pf), L.ptr(out), np_, L.ptr(arg), 32, 32, np_, lp, 2,
                                         lp * 128, np_ * lp * 128, L.ptr(qa), L.ptr(pa), None, st))
tot = {k: 0.0 for k in cases}; names = {}
for rep in range(5):
    for k in cases:
        for _ in range(10): call(k)
        torch.cuda.synchronize(); names[k] = lib.evdr_last_fwd_kernel().decode()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40): call(k)
        b.record(); torch.cuda.synchronize(); tot[k] += a.elapsed_time(b) / 40 / 5 * 1e3
print("lp   argmax   125 pages   500 pages   per page of a workgroup   fixed     MFMA floor per page @2.1 GHz   kernel")
for lp in (96, 192, 206, 224, 384):
    for am in (False, True):
        t1, t4 = tot[(125, lp, am)], tot[(500, lp, am)]
        per = (t4 - t1) / 3
        halves = sum(1 for h in range(2 * ((lp + 31) // 32)) if h * 16 < lp)
        floor = halves * 48 * 16 * 2 / 2.1e3
        print(f"{lp:4d}   {int(am)}      {t1:8.1f}    {t4:8.1f}    {per:8.2f}              {t1 - per:6.1f}    {floor:6.2f}                          {names[(500, lp, am)]}", flush=True)
