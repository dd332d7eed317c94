"""Round 3: tests/test_gpu_fullsize.py::test_config2 failed ONCE on token additivity (|sa + sb - s1| = 8.7e-3 on one box, after
two green runs of the same sources).  This probe repeats that property (and plain determinism) many times in one process and
reports WHERE the scores differ: which launches, which (query, page) entries, by how much, against a fixed first result.
usage: python scratch/additivity_probe.py [reps]"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import evdr_amd  # noqa: F401
from evdr_amd.corpus import PageCorpus
from evdr_amd import _lib as L
import test_gpu_fullsize as T
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n, nq = 6847, 256
P, Q, tgt = T.synth(n, nq, dev, seed=12)
corpus = PageCorpus.from_tensor(P)
ma = torch.zeros(nq, T.LQ, dtype=torch.bool, device=dev); ma[:, ::2] = True
mb = ~ma
s1 = corpus.score(Q).clone(); k1 = L.load().evdr_last_fwd_kernel().decode()
sa0 = corpus.score(Q, ma).clone(); ka = L.load().evdr_last_fwd_kernel().decode()
sb0 = corpus.score(Q, mb).clone()
print("kernels:", k1, "|", ka)
print("first: |sa+sb-s1| max", (sa0 + sb0 - s1).abs().max().item())
bad = 0
for r in range(reps):
    s = corpus.score(Q); a = corpus.score(Q, ma); b = corpus.score(Q, mb)
    torch.cuda.synchronize()
    for name, got, ref in (("s1", s, s1), ("sa", a, sa0), ("sb", b, sb0)):
        d = (got - ref).abs()
        if d.max().item() != 0:
            bad += 1
            idx = (d > 0).nonzero()
            qs, ps = idx[:, 0].unique().tolist(), idx[:, 1].unique().tolist()
            print(f"rep {r} {name}: {len(idx)} entries differ, max {d.max().item():.3e}; queries {qs[:12]}{'...' if len(qs) > 12 else ''} "
                  f"pages {ps[:12]}{'...' if len(ps) > 12 else ''}", flush=True)
    add = (a + b - s).abs().max().item()
    if add >= 2e-5:
        print(f"rep {r}: additivity {add:.3e}", flush=True)
print(f"done: {reps} reps, {bad} differing launches")
