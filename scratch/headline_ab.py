"""Headline kernel time (1024 queries x N pages, bf16) with the library given by EVDR_LIB_AB (A/B of two builds on one box)."""
import os, sys, torch
sys.path.insert(0, "."); import evdr_amd
from evdr_amd import _lib
if os.environ.get("EVDR_LIB_AB"): _lib.LIB_PATH = os.environ["EVDR_LIB_AB"]
import bench as B
from evdr_amd.corpus import PageCorpus
dev = torch.device("cuda:0"); pages = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
P = B.gen_pages(0, pages, dev); corpus = PageCorpus.from_tensor(P, None)
Q, _ = B.make_queries(1024, pages, P, 0, pages, dev, 1)
out = torch.empty((1024, pages), dtype=torch.float32, device=dev)
corpus.score(Q, None, out=out); torch.cuda.synchronize()
ts = []
for _ in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); corpus.score(Q, None, out=out); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ms = sum(ts) / len(ts)
print(f"{os.environ.get('EVDR_LIB_AB','default'):28s} {ms:8.2f} ms  {1024*pages*B.FLOP_PER_PAIR/ms/1e9:7.1f} TFLOP/s  (min {min(ts):.2f} max {max(ts):.2f})  chk {float(out.double().sum()):.4f}", flush=True)
