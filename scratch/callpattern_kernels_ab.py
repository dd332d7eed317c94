"""The C-ABI entry points the reference's CALL PATTERN issues per step, timed per entry over several BUILDS of the library (interleaved)."""
import ctypes as C, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import evdr_amd
from evdr_amd import _lib as L, ops
paths = [L.LIB_PATH if p == "default" else os.path.abspath(p) for p in sys.argv[1:]]
handles = []
for p in paths:
    lib = C.CDLL(p)
    for name, (res, args) in L.SIGNATURES.items():
        fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    handles.append(lib)
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(3)
B, N, Ls, Lt = 32, 500, 206, 1030
x = torch.randn(N, Ls, 128, device=dev, generator=g); pm = torch.ones(N, Ls, dtype=torch.bool, device=dev)
Q = torch.nn.functional.normalize(torch.randn(B, 32, 128, device=dev, generator=g), dim=-1); qm = torch.ones(B, 32, dtype=torch.bool, device=dev)
ss = torch.randn(B, N, device=dev, generator=g) * 3; st = torch.randn(B, N, device=dev, generator=g) * 3
gy = torch.randn(N, Ls, 128, device=dev, generator=g) * 1e-3
ea, es = torch.zeros_like(x), torch.zeros_like(x)
WS = ops.infonce_workspace(B, dev)
def jobs():
    y, norm, planes, amax = ops.l2norm_forward(x, pm, 1e-12, want_planes=True)
    tm, pf = ops.pack_pmask(pm, N, Ls, dev)
    qp, qa = ops.split_f32(Q)
    out, arg = ops.maxsim_forward_prepared(qp, qa, planes, amax, qm, tm, pf, want_argmax=True)
    gs = torch.randn(B, N, device=dev, generator=g) * 1e-2
    return {
        "l2norm_fwd_split": lambda: ops.l2norm_forward(x, pm, 1e-12, want_planes=True),
        "pack_pmask+flag": lambda: (ops.pack_pmask(pm, N, Ls, dev), ops.flag_nonfinite(planes[0], pm, pf)),
        "split_f32(Q)": lambda: ops.split_f32(Q),
        "student fwd+argmax": lambda: ops.maxsim_forward_prepared(qp, qa, planes, amax, qm, tm, pf, want_argmax=True),
        "infonce (two-launch form)": lambda: ops.infonce_distill(ss, st, 0.1, want_grad=True),
        "infonce (one-launch form)": lambda: ops.infonce_distill(ss, st, 0.1, want_grad=True, ws=WS),
        "maxsim_bwd (dP)": lambda: ops.maxsim_backward(gs, Q, qm, pm, arg, N, Ls),
        "l2norm_bwd": lambda: ops.l2norm_backward(gy, x, pm, norm, 1e-12),
        "adamw_step": lambda: ops.adamw_step(gy, x, ea, es, 1e-3, (0.9, 0.999), 1e-8, 1e-2, 1),
    }
res = {}
for rnd in range(6):
    for p, lib in zip(paths, handles):
        L._lib = lib
        for name, fn in jobs().items():
            for _ in range(5): fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30): fn()
            b.record(); torch.cuda.synchronize()
            if rnd: res.setdefault((name, os.path.basename(p)), []).append(a.elapsed_time(b) / 30 * 1e3)
names = []
for (n, _p) in res:
    if n not in names: names.append(n)
for n in names:
    print(f"{n:28s} " + "   ".join(f"{os.path.basename(p)}: {sum(res[(n, os.path.basename(p))]) / 5:7.1f} us" for p in paths), flush=True)
