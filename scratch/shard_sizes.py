"""Per-rank efficiency of the headline kernel at the shard sizes of 1/2/4/8-GPU strong scaling (1024 queries)."""
import _hooks as H
import os, sys, torch
sys.path.insert(0, "."); import evdr_amd, bench as B
from evdr_amd.corpus import PageCorpus
dev = torch.device("cuda:0")
Pall = B.gen_pages(0, 50000, dev)
Q, _ = B.make_queries(1024, 50000, Pall, 0, 50000, dev, 1)
for pages in (12500, 25000, 50000):
    corpus = PageCorpus.from_tensor(Pall[:pages], None)
    out = torch.empty((1024, pages), dtype=torch.float32, device=dev)
    line = f"pages={pages:6d}"
    for ppb in sys.argv[1].split(","):
        H.set_ppb(ppb or 0)
        corpus.score(Q, None, out=out); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); corpus.score(Q, None, out=out); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        ms = min(ts)
        line += f" | ppb={ppb:>3s}: {ms:8.2f} ms {1024*pages*B.FLOP_PER_PAIR/ms/1e9:7.1f} TF"
    print(line, flush=True)
    ts, ti = corpus.topk(Q, None, 100); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); corpus.topk(Q, None, 100); b.record(); torch.cuda.synchronize()
    print(f"             score+topk {a.elapsed_time(b):8.2f} ms", flush=True)
