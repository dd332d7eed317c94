"""Round 3: hunting a rare score flake (one token-additivity failure in tests/test_gpu_fullsize.py, consistent with ONE
(token, page) maximum missing its best patch in one launch).  Repeats forward launches of several shapes and kernel instances
and compares every output with the first one of its shape BIT FOR BIT; any difference is a data hazard inside a launch (the
kernels have no atomics and no run-to-run freedom for lq <= 32).
usage: python scratch/determinism_stress.py <seconds per shape> [variant]"""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import evdr_amd  # noqa: F401
from evdr_amd.corpus import PageCorpus
from evdr_amd import _lib as L
import test_gpu_fullsize as T
dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = L.load()
lib.evdr_debug_set_fwd_variant(variant)
P, Qall, _ = T.synth(20000, 1024, dev, seed=12)
shapes = [("bf16 nq=256 pages=6847", 256, 6847, False), ("bf16 nq=1024 pages=20000", 1024, 20000, False), ("bf16 nq=64 pages=20000", 64, 20000, False),
          ("bf16 nq=8 pages=20000", 8, 20000, False), ("bf16 nq=1 pages=20000", 1, 20000, False), ("bf16 nq=20 pages=20000", 20, 20000, False),
          ("fp32 nq=32 pages=500", 32, 500, True), ("fp32 nq=256 pages=2000", 256, 2000, True)]
total_bad = 0
masked = len(sys.argv) > 3 and sys.argv[3] == "masked"        # query masks (every other token / random) in front of every launch
for name, nq, npg, f32 in shapes:
    Pp = P[:npg].float() if f32 else P[:npg]
    Q = Qall[:nq].float() if f32 else Qall[:nq].contiguous()
    corpus = PageCorpus.from_tensor(Pp)
    if masked:
        qm_all = [torch.zeros(nq, T.LQ, dtype=torch.bool, device=dev) for _ in range(2)]
        qm_all[0][:, ::2] = True
        qm_all[1] = ~qm_all[0]
        refs = [corpus.score(Q, m).clone() for m in qm_all]
        full = corpus.score(Q).clone()
        kern = lib.evdr_last_fwd_kernel().decode()
        t0, n, bad = time.time(), 0, 0
        while time.time() - t0 < budget:
            for _ in range(4):
                a = corpus.score(Q, qm_all[0]); b = corpus.score(Q, ~qm_all[0])          # like the test: a temporary mask tensor
                n += 2
                da, db = int((a != refs[0]).sum().item()), int((b != refs[1]).sum().item())
                add = (a + b - full).abs().max().item()
                if da or db or add >= 2e-5:
                    bad += 1
                    print(f"  {name}: launch {n}: sa differs in {da}, sb in {db} entries; additivity {add:.3e}", flush=True)
        total_bad += bad
        print(f"{name} [{kern}] masked: {n} launches, {bad} bad", flush=True)
        continue
    ref = corpus.score(Q).clone()
    kern = lib.evdr_last_fwd_kernel().decode()
    out = torch.empty_like(ref)
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        for _ in range(8):
            corpus.score(Q, out=out)
            d = (out != ref)
            cnt = int(d.sum().item())
            n += 1
            if cnt:
                bad += 1
                idx = d.nonzero()
                mx = (out - ref).abs().max().item()
                print(f"  {name}: launch {n}: {cnt} entries differ (max {mx:.3e}); first (q,p) {idx[:6].tolist()}", flush=True)
    total_bad += bad
    print(f"{name} [{kern}] variant {variant}: {n} launches, {bad} with differences", flush=True)
print(f"done: {total_bad} differing launches")
