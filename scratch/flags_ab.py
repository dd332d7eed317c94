"""the barrier-free (flag) stage hand-over against the shipped s_barrier hand-over, experiment build,
interleaved in one process.  Variant 70 = student forward <2,2,true,3,...>+flags, variant 71 = headline <4,1,false,8,...>+flags;
variant 0 = the shipped instances.  Scores (and arg-max) must be bit-identical; the poll-timeout word must stay 0.
  python scratch/flags_ab.py [student|headline|all]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd, bench as B
from evdr_amd import _lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.PKG_DIR), "scratch", "_variants", "libevdr_exp.so")
from evdr_amd import ops
from evdr_amd.corpus import PageCorpus
what = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = torch.device("cuda:0"); torch.manual_seed(0); lib = L.load()
lib.evdr_experiment_set_dbg_buffer.argtypes = [ctypes.c_void_p]; lib.evdr_experiment_set_dbg_buffer.restype = None
dbg = torch.zeros((4096, 8, 8), dtype=torch.int64, device=dev); lib.evdr_experiment_set_dbg_buffer(dbg.data_ptr())
def unit(*s): return torch.nn.functional.normalize(torch.randn(*s, device=dev), dim=-1)

if what in ("student", "all"):
    for nq, np_, lp in [(32, 500, 206), (32, 125, 206), (64, 500, 206), (32, 500, 192)]:
        Q, P = unit(nq, 32, 128), unit(np_, lp, 128)
        qp, qa = ops.split_f32(Q); pp, pa = ops.split_f32(P)
        tm, pf = ops.pack_pmask(None, np_, lp, dev)
        out = torch.empty(nq, np_, device=dev); arg = torch.empty(nq, np_, 32, dtype=torch.int16, device=dev)
        st = L.current_stream_handle(dev)
        def call():
            L.check(lib.evdr_maxsim_fwd_prepared(L.ptr(qp), L.ptr(pp), None, L.ptr(tm), L.ptr(pf), L.ptr(out), np_, L.ptr(arg), nq, 32, np_, lp, 2,
                                                 lp * 128, np_ * lp * 128, L.ptr(qa), L.ptr(pa), None, st))
        variants = (0, 70, 72)
        tot = {v: 0.0 for v in variants}; ref = None; names = {}
        for rep in range(6):
            for v in variants:
                lib.evdr_debug_set_fwd_variant(v)
                out.zero_(); arg.zero_()
                for _ in range(20): call()
                torch.cuda.synchronize()
                names[v] = lib.evdr_last_fwd_kernel().decode()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(50): call()
                b.record(); torch.cuda.synchronize(); tot[v] += a.elapsed_time(b) / 50
                if ref is None: ref = (out.clone(), arg.clone())
                same = torch.equal(ref[0], out) and torch.equal(ref[1], arg)
                if not same:
                    print(f"  variant {v} DIFFERS: scores max |d| {(ref[0] - out).abs().max().item():.3e}, arg-max mismatches {(ref[1] != arg).sum().item()}", flush=True)
        lib.evdr_debug_set_fwd_variant(0)
        for v in variants:
            print(f"student nq={nq:3d} np={np_:4d} lp={lp}  variant {v:2d} {names[v]:70s} {tot[v] / 6 * 1e3:8.1f} us", flush=True)
        print(f"  poll-timeout word: {int(dbg[0, 0, 0].item()):#x}", flush=True)

if what in ("headline", "all"):
    for nq, pages, rounds in [(1024, 20000, 6), (256, 20000, 6), (32, 20000, 6)]:
        P = B.gen_pages(0, pages, dev); corpus = PageCorpus.from_tensor(P, None)
        Q, _ = B.make_queries(max(nq, 32), pages, P, 0, pages, dev, 1); Q = Q[:nq].contiguous()
        out = torch.empty((nq, pages), dtype=torch.float32, device=dev); variants = (0, 71, 73); res = {v: [] for v in variants}; ref = None; names = {}
        for rnd in range(rounds + 1):
            for v in variants:
                lib.evdr_debug_set_fwd_variant(v)
                out.zero_()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); corpus.score(Q, None, out=out); b.record(); torch.cuda.synchronize()
                names[v] = lib.evdr_last_fwd_kernel().decode()
                if ref is None: ref = out.clone()
                if not torch.equal(out, ref):
                    print(f"  variant {v} DIFFERS: max |d| {(out - ref).abs().max().item():.3e}, entries {(out != ref).sum().item()}", flush=True)
                if rnd > 0: res[v].append(a.elapsed_time(b))
        lib.evdr_debug_set_fwd_variant(0)
        for v in variants:
            ts = res[v]; print(f"headline nq={nq} pages={pages} variant {v:3d} {names[v]:66s} mean {sum(ts)/len(ts):8.4f} ms  min {min(ts):8.4f}", flush=True)
        print(f"  poll-timeout word: {int(dbg[0, 0, 0].item()):#x}", flush=True)
        del P, corpus
