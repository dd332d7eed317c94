"""Whole training STEP (bench_train.time_mode) over several BUILDS of the library, interleaved in one process: the package's library
handle is swapped between rounds.  usage: python scratch/step_lib_ab.py <rounds> <mode,mode,...> <lib.so> [...]  ("default" = the package's)"""
import ctypes as C, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import evdr_amd
from evdr_amd import _lib as L
import bench_train as BT
rounds = int(sys.argv[1]); modes = sys.argv[2].split(",")
paths = [L.LIB_PATH if p == "default" else os.path.abspath(p) for p in sys.argv[3:]]
handles = []
for p in paths:
    lib = C.CDLL(p)
    for name, (res, args) in L.SIGNATURES.items():
        fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    assert lib.evdr_version() == L.ABI_VERSION
    handles.append(lib)
dev = torch.device("cuda:0")
inp = BT.make_inputs(500, 32, dev)
res = {}
for rnd in range(rounds + 1):
    for p, lib in zip(paths, handles):
        L._lib = lib                                   # every ops.* call goes through L.load(), which returns this handle
        for mode in modes:
            r = BT.time_mode(inp, mode, steps=128, warmup=32)
            if rnd: res.setdefault((os.path.basename(p), mode), []).append(r["ms_per_step"] * 1e3)
for mode in modes:
    print(f"{mode:22s} " + "   ".join(f"{os.path.basename(p)}: {sum(res[(os.path.basename(p), mode)]) / rounds:7.1f} us (min {min(res[(os.path.basename(p), mode)]):7.1f})" for p in paths), flush=True)
