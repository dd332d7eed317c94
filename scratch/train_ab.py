"""Kernel-only A/B of the training-step kernels over several BUILDS of the library (one process, interleaved):
teacher forward (32 x 500 x 1030, fp16 hi/lo planes), student forward + argmax (32 x 500 x 206), fused backward + AdamW.
usage: python scratch/train_ab.py <rounds> <lib.so> [...]   ("default" = the package's)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
from evdr_amd import _lib as L, ops
rounds = int(sys.argv[1]); paths = [L.LIB_PATH if p == "default" else os.path.abspath(p) for p in sys.argv[2:]]
dev = torch.device("cuda:0"); torch.manual_seed(0)
def unit(*s): return torch.nn.functional.normalize(torch.randn(*s, device=dev), dim=-1)
B, N, Lt, Ls = 32, 500, 1030, 206
Q, Pt, Ps = unit(B, 32, 128), unit(N, Lt, 128), unit(N, Ls, 128)
qp, qa = ops.split_f32(Q); tp, ta = ops.split_f32(Pt); sp, sa = ops.split_f32(Ps)
ttm, tpf = ops.pack_pmask(None, N, Lt, dev); stm, spf = ops.pack_pmask(None, N, Ls, dev)
out = torch.empty(B, N, device=dev); arg = torch.empty(B, N, 32, dtype=torch.int16, device=dev)
g = torch.randn(B, N, device=dev) * 1e-2
x = Ps.clone(); ea = torch.zeros_like(x); es = torch.zeros_like(x)
st = torch.cuda.current_stream(dev).cuda_stream
libs = []
for p in paths:
    lib = C.CDLL(p)
    for n in ("evdr_maxsim_fwd_prepared", "evdr_maxsim_bwd_adamw"):
        getattr(lib, n).restype = C.c_int; getattr(lib, n).argtypes = L.SIGNATURES[n][1]
    libs.append(lib)
def teacher(lib): assert lib.evdr_maxsim_fwd_prepared(qp.data_ptr(), tp.data_ptr(), None, ttm.data_ptr(), tpf.data_ptr(), out.data_ptr(), N, None, B, 32, N, Lt, 2, Lt * 128, N * Lt * 128, qa.data_ptr(), ta.data_ptr(), None, st) == 0
def student(lib): assert lib.evdr_maxsim_fwd_prepared(qp.data_ptr(), sp.data_ptr(), None, stm.data_ptr(), spf.data_ptr(), out.data_ptr(), N, arg.data_ptr(), B, 32, N, Ls, 2, Ls * 128, N * Ls * 128, qa.data_ptr(), sa.data_ptr(), None, st) == 0
def bwd(lib): assert lib.evdr_maxsim_bwd_adamw(g.data_ptr(), Q.data_ptr(), None, None, arg.data_ptr(), x.data_ptr(), ea.data_ptr(), es.data_ptr(), B, 32, N, Ls, 128, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 1, 1e-12, None, st) == 0
res = {}
for name, fn, n in (("teacher_fwd", teacher, 20), ("student_fwd_argmax", student, 50), ("bwd_adamw", bwd, 50)):
    student(libs[0]); ref = None
    for rnd in range(rounds + 1):
        for p, lib in zip(paths, libs):
            for _ in range(5): fn(lib)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n): fn(lib)
            b.record(); torch.cuda.synchronize()
            if name != "bwd_adamw":
                if ref is None: ref = (out.clone(), arg.clone())
                else: assert torch.equal(ref[0], out) and (name == "teacher_fwd" or torch.equal(ref[1], arg)), (name, p)
            if rnd: res.setdefault((name, p), []).append(a.elapsed_time(b) / n * 1e3)
    for p in paths:
        ts = res[(name, p)]
        print(f"{name:20s} {os.path.basename(p):28s} mean {sum(ts)/len(ts):8.1f} us  min {min(ts):8.1f} us", flush=True)
