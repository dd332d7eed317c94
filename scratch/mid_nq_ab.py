"""A/B of forward variants at mid query counts (1..8 query groups per page chunk)."""
import _hooks as H
import os, sys, torch
sys.path.insert(0, "."); import evdr_amd, bench as B
from evdr_amd.corpus import PageCorpus
dev = torch.device("cuda:0"); pages = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
variants = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,2".split(","))]
P = B.gen_pages(0, pages, dev); corpus = PageCorpus.from_tensor(P, None)
Qall, _ = B.make_queries(1024, pages, P, 0, pages, dev, 1)
ref = {}
for nq in (32, 64, 128, 256, 1024):
    Q = Qall[:nq].contiguous(); out = torch.empty((nq, pages), dtype=torch.float32, device=dev)
    line = f"nq={nq:4d}"
    for v in variants:
        H.set_variant(v)
        corpus.score(Q, None, out=out); torch.cuda.synchronize()
        if nq not in ref: ref[nq] = out.clone()
        same = bool(torch.equal(ref[nq], out))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5 if nq <= 128 else 2
        a.record()
        for _ in range(reps): corpus.score(Q, None, out=out)
        b.record(); torch.cuda.synchronize(); ms = a.elapsed_time(b) / reps
        line += f" | v{v}: {ms:8.3f} ms {nq*pages*B.FLOP_PER_PAIR/ms/1e9:7.1f} TF {'ok' if same else 'DIFF'}"
    print(line, flush=True)
