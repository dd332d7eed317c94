"""Interleaved A/B of several BUILDS of the library on the training step's two forwards and a headline-like launch.
usage: python scratch/fwd_store_ab.py <lib.so> [<lib.so> ...]   ("default" = the package's)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, "."); import evdr_amd
from evdr_amd import _lib as L, ops
paths = [L.LIB_PATH if p == "default" else os.path.abspath(p) for p in sys.argv[1:]]
dev = torch.device("cuda:0")
libs = []
for p in paths:
    lib = C.CDLL(p)
    lib.evdr_maxsim_fwd_prepared.restype = C.c_int
    lib.evdr_maxsim_fwd_prepared.argtypes = L.SIGNATURES["evdr_maxsim_fwd_prepared"][1]
    libs.append(lib)
stream = torch.cuda.current_stream(dev).cuda_stream
ONLY = os.environ.get("AB_ONLY", "")
def unit(*s, g): return torch.nn.functional.normalize(torch.randn(*s, device=dev, generator=g), dim=-1)
for name, (nq, npg, lp, f32, argmax, reps) in {"student fwd + argmax 32x500x206 f32": (32, 500, 206, True, True, 40), "teacher fwd 32x500x1030 f32": (32, 500, 1030, True, False, 20),
                                               "student shard 32x63x206 f32": (32, 63, 206, True, True, 40),
                                               "headline-like 1024x4000x1030 bf16": (1024, 4000, 1030, False, False, 5), "32x20000x1030 bf16": (32, 20000, 1030, False, False, 10)}.items():
    if ONLY and ONLY not in name: continue
    g = torch.Generator(device=dev).manual_seed(7)
    Q, P = unit(nq, 32, 128, g=g), unit(npg, lp, 128, g=g)
    if f32:
        planes, pamax = ops.split_f32(P); qpl, qamax = ops.split_f32(Q); npl = 2
    else:
        planes, pamax, qpl, qamax, npl = P.bfloat16()[None].contiguous(), None, Q.bfloat16()[None].contiguous(), None, 1
    tm, pf = ops.pack_pmask(None, npg, lp, dev)
    out = torch.empty((nq, npg), dtype=torch.float32, device=dev)
    arg = torch.empty((nq, npg, 32), dtype=torch.int16, device=dev) if argmax else None
    def run(lib):
        rc = lib.evdr_maxsim_fwd_prepared(qpl.data_ptr(), planes.data_ptr(), None, tm.data_ptr(), pf.data_ptr(), out.data_ptr(), npg, arg.data_ptr() if argmax else None,
                                          nq, 32, npg, lp, npl, planes.stride(1), planes.stride(0), qamax.data_ptr() if f32 else None, pamax.data_ptr() if f32 else None, None, stream)
        assert rc == 0, rc
    res = {p: [] for p in paths}; ref = None; same = {}
    for rnd in range(6):
        for p, lib in zip(paths, libs):
            for _ in range(3): run(lib)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps): run(lib)
            b.record(); torch.cuda.synchronize()
            if rnd == 0:
                if ref is None: ref = (out.clone(), arg.clone() if argmax else None)
                else: same[p] = bool(torch.equal(out, ref[0]) and (not argmax or torch.equal(arg, ref[1])))
            else: res[p].append(a.elapsed_time(b) / reps * 1e3)
    print(name + ":  " + "   ".join(f"{os.path.basename(p)} {sum(res[p]) / len(res[p]):9.1f} us (min {min(res[p]):9.1f}){'' if p == paths[0] else ' same bits ' + str(same[p])}" for p in paths), flush=True)
