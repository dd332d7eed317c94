"""Fused update (evdr_maxsim_bwd_adamw_planes as the training step launches it) and the dP kernel of several BUILDS of the library, interleaved in
one process; results must be bit-identical to the first library's.  usage: python scratch/bwd_lib_ab.py default scratch/ab/libevdr_bw64.so ..."""
import ctypes as C, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import evdr_amd
from evdr_amd import _lib as L
dev = torch.device("cuda:0")
paths = [L.LIB_PATH if p == "default" else os.path.abspath(p) for p in sys.argv[1:]]
libs = {}
for p in paths:
    lib = C.CDLL(p)
    for name in ("evdr_maxsim_bwd", "evdr_maxsim_bwd_adamw_planes"):
        getattr(lib, name).restype, getattr(lib, name).argtypes = L.SIGNATURES[name]
    libs[os.path.basename(p)] = lib
print(L.SIGNATURES["evdr_maxsim_bwd_adamw_planes"][1].__len__(), "arguments")
st = torch.cuda.current_stream(dev).cuda_stream
g0 = torch.Generator(device=dev).manual_seed(3)
for nq, lq, npg, lp, hot in ((32, 32, 500, 206, 0.0), (32, 32, 500, 206, 0.4), (32, 32, 63, 206, 0.0), (32, 32, 125, 206, 0.0), (32, 32, 250, 206, 0.0), (32, 32, 375, 206, 0.0),
                              (32, 32, 500, 103, 0.0), (32, 32, 125, 103, 0.0), (32, 32, 250, 103, 0.0), (64, 32, 500, 206, 0.0), (32, 32, 100, 1030, 0.0), (32, 32, 30, 1030, 0.0)):
    Q = torch.randn(nq, lq, 128, device=dev, generator=g0); gr = torch.randn(nq, npg, device=dev, generator=g0) * 1e-2
    arg = torch.randint(0, lp, (nq, npg, lq), device=dev, generator=g0)
    if hot > 0: arg[torch.rand(nq, npg, lq, device=dev, generator=g0) < hot] = 17
    arg = arg.to(torch.int16)
    x0 = torch.randn(npg, lp, 128, device=dev, generator=g0)
    pm = torch.ones(npg, lp, dtype=torch.uint8, device=dev)
    state = {}
    for tag in libs:
        state[tag] = dict(x=x0.clone(), ea=torch.zeros_like(x0), es=torch.zeros_like(x0), dP=torch.empty_like(x0),
                          planes=torch.empty(2, npg, lp, 128, dtype=torch.float16, device=dev), amax=torch.zeros(1, dtype=torch.int32, device=dev),
                          pf=torch.zeros(npg, dtype=torch.int32, device=dev))
    res = {}
    def call(tag, kind):
        lib, s = libs[tag], state[tag]
        if kind == "bwd":
            rc = lib.evdr_maxsim_bwd(gr.data_ptr(), Q.data_ptr(), None, pm.data_ptr(), arg.data_ptr(), s["dP"].data_ptr(), nq, lq, npg, lp, 128, st)
        else:
            rc = lib.evdr_maxsim_bwd_adamw_planes(gr.data_ptr(), Q.data_ptr(), None, pm.data_ptr(), arg.data_ptr(), s["x"].data_ptr(), s["ea"].data_ptr(), s["es"].data_ptr(),
                                                  nq, lq, npg, lp, 128, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 1, 1e-12, None, s["planes"].data_ptr(), s["amax"].data_ptr(),
                                                  s["pf"].data_ptr(), st)
        assert rc == 0, rc
    for rnd in range(5):
        for tag in libs:
            for kind in ("bwd", "adamw"):
                for _ in range(5): call(tag, kind)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(30): call(tag, kind)
                b.record(); torch.cuda.synchronize()
                res.setdefault((tag, kind), []).append(a.elapsed_time(b) / 30 * 1e3)
    first = next(iter(libs))
    same = all(torch.equal(state[t]["dP"], state[first]["dP"]) and torch.equal(state[t]["x"], state[first]["x"]) and torch.equal(state[t]["planes"], state[first]["planes"]) for t in libs)
    f = lambda k: sum(res[k]) / len(res[k])
    print(f"nq={nq:3d} np={npg:4d} lp={lp:4d} hot={hot:.1f} bit-identical={same}: " + "   ".join(f"{t}: dP {f((t, 'bwd')):6.1f} us  fused update {f((t, 'adamw')):6.1f} us" for t in libs), flush=True)
