"""A/B of forward variants in the HBM-bound regime (few queries per pass)."""
import _hooks as H
import os, sys, torch
sys.path.insert(0, "."); import evdr_amd, bench as B
from evdr_amd.corpus import PageCorpus
dev = torch.device("cuda:0"); pages = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
variants = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1".split(","))]
P = B.gen_pages(0, pages, dev); corpus = PageCorpus.from_tensor(P, None)
Qall, _ = B.make_queries(64, pages, P, 0, pages, dev, 1)
ref = {}
for nq in (1, 2, 4, 8, 16, 32):
    Q = Qall[:nq].contiguous(); out = torch.empty((nq, pages), dtype=torch.float32, device=dev)
    line = f"nq={nq:3d}"
    for v in variants:
        H.set_variant(v)
        corpus.score(Q, None, out=out); torch.cuda.synchronize()
        if nq not in ref: ref[nq] = out.clone()
        same = bool(torch.equal(ref[nq], out))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): corpus.score(Q, None, out=out)
        b.record(); torch.cuda.synchronize(); ms = a.elapsed_time(b) / 5
        line += f" | v{v}: {ms:7.3f} ms {pages*263680/ms/1e9:5.2f} TB/s {'ok' if same else 'DIFF'}"
    print(line, flush=True)
