"""Headline shape (1024 queries x 20 000 pages x 1030 patches, bf16): kernel time against pages per workgroup (debug hook)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd, bench as B
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _hooks as H
from evdr_amd.corpus import PageCorpus
dev = torch.device("cuda:0"); pages = 20000
P = B.gen_pages(0, pages, dev); corpus = PageCorpus.from_tensor(P, None)
Q, _ = B.make_queries(1024, pages, P, 0, pages, dev, 1)
out = torch.empty((1024, pages), dtype=torch.float32, device=dev)
def t(n=6):
    corpus.score(Q, None, out=out); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): corpus.score(Q, None, out=out)
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
for rnd in range(2):
    for ppb in (0, 16, 24, 32, 48, 64, 80):
        H.set_ppb(ppb); ms = t()
        print(f"round {rnd} pages per workgroup {ppb if ppb else 'default'}: {ms:8.3f} ms  {1024*pages*B.FLOP_PER_PAIR/ms/1e9:7.1f} TFLOP/s", flush=True)
H.set_ppb(0)
