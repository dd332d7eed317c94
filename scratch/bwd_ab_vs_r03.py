"""Cost of the bit-reproducible backward: evdr_maxsim_bwd (dP to HBM) and evdr_maxsim_bwd_adamw of this round's library against the
round-3 library (scratch/ab/libevdr_r03.so), interleaved in one process; uniform arg-max and one salient patch (heavy shared row)."""
import ctypes as C, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import evdr_amd
from evdr_amd import _lib as L
dev = torch.device("cuda:0")
libs = {}
for tag, p in (("r03", os.path.join(R, "scratch", "ab", "libevdr_r03.so")), ("r04", L.LIB_PATH)):
    lib = C.CDLL(p)
    for name in ("evdr_maxsim_bwd", "evdr_maxsim_bwd_adamw"):
        getattr(lib, name).restype, getattr(lib, name).argtypes = L.SIGNATURES[name]
    libs[tag] = lib
st = torch.cuda.current_stream(dev).cuda_stream
g0 = torch.Generator(device=dev).manual_seed(3)
for nq, lq, npg, lp, hot in ((32, 32, 500, 206, 0.0), (32, 32, 500, 206, 0.4), (32, 32, 500, 1030, 0.0), (64, 32, 500, 206, 0.0), (256, 32, 128, 206, 0.0), (32, 32, 2000, 206, 0.3)):
    Q = torch.randn(nq, lq, 128, device=dev, generator=g0); gr = torch.randn(nq, npg, device=dev, generator=g0) * 1e-2
    arg = torch.randint(0, lp, (nq, npg, lq), device=dev, generator=g0)
    if hot > 0: arg[torch.rand(nq, npg, lq, device=dev, generator=g0) < hot] = 17
    arg = arg.to(torch.int16)
    dP = torch.empty(npg, lp, 128, device=dev); x = torch.randn(npg, lp, 128, device=dev, generator=g0); ea = torch.zeros_like(x); es = torch.zeros_like(x)
    res = {}
    for rnd in range(5):
        for tag, lib in libs.items():
            for kind in ("bwd", "adamw"):
                def call():
                    if kind == "bwd":
                        rc = lib.evdr_maxsim_bwd(gr.data_ptr(), Q.data_ptr(), None, None, arg.data_ptr(), dP.data_ptr(), nq, lq, npg, lp, 128, st)
                    else:
                        rc = lib.evdr_maxsim_bwd_adamw(gr.data_ptr(), Q.data_ptr(), None, None, arg.data_ptr(), x.data_ptr(), ea.data_ptr(), es.data_ptr(), nq, lq, npg, lp, 128,
                                                       1e-3, 0.9, 0.999, 1e-8, 1e-2, 1, 1e-12, None, st)
                    assert rc == 0
                for _ in range(5): call()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(30): call()
                b.record(); torch.cuda.synchronize()
                res.setdefault((tag, kind), []).append(a.elapsed_time(b) / 30 * 1e3)
    f = lambda k: sum(res[k]) / len(res[k])
    print(f"nq={nq:3d} np={npg:4d} lp={lp:4d} hot={hot:.1f}:  dP kernel r03 {f(('r03','bwd')):7.1f} us  r04 {f(('r04','bwd')):7.1f} us ({f(('r04','bwd'))/f(('r03','bwd'))-1:+.1%})   "
          f"fused update r03 {f(('r03','adamw')):7.1f} us  r04 {f(('r04','adamw')):7.1f} us ({f(('r04','adamw'))/f(('r03','adamw'))-1:+.1%})", flush=True)
