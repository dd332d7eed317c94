"""Student / teacher forward with 3-tile vs 4-tile stages (debug variants 10 / 11) on the product library, interleaved."""
import os, sys, torch
sys.path.insert(0, "."); import evdr_amd
from evdr_amd import _lib as L, ops
dev = torch.device("cuda:0"); lib = L.load()
def unit(*s, g): return torch.nn.functional.normalize(torch.randn(*s, device=dev, generator=g), dim=-1)
for name, (nq, npg, lp, argmax) in {"student 32x500x206 argmax": (32, 500, 206, True), "student 32x500x206": (32, 500, 206, False), "32x500x224 argmax": (32, 500, 224, True),
                                    "teacher 32x500x1030": (32, 500, 1030, False)}.items():
    g = torch.Generator(device=dev).manual_seed(7)
    Q, P = unit(nq, 32, 128, g=g), unit(npg, lp, 128, g=g)
    planes, pamax = ops.split_f32(P); qpl, qamax = ops.split_f32(Q)
    tm, pf = ops.pack_pmask(None, npg, lp, dev)
    res = {}; ref = None
    for rnd in range(6):
        for var in (0, 10, 11):
            lib.evdr_debug_set_fwd_variant(var)
            for _ in range(3): out, arg = ops.maxsim_forward_prepared(qpl, qamax, planes, pamax, None, tm, pf, want_argmax=argmax)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(40): out, arg = ops.maxsim_forward_prepared(qpl, qamax, planes, pamax, None, tm, pf, want_argmax=argmax)
            b.record(); torch.cuda.synchronize()
            kern = lib.evdr_last_fwd_kernel().decode()
            if ref is None: ref = out.clone()
            assert torch.equal(out, ref)
            if rnd: res.setdefault((var, kern), []).append(a.elapsed_time(b) / 40 * 1e3)
    lib.evdr_debug_set_fwd_variant(0)
    print(name + ": " + "   ".join(f"variant {v} {k[20:46]} {sum(t)/len(t):7.1f} us (min {min(t):7.1f})" for (v, k), t in res.items()), flush=True)
