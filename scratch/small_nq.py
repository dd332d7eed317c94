"""Latency regime: few queries per pass -> the kernel should be bound by streaming the corpus from HBM (263 680 B/page)."""
import sys, torch
sys.path.insert(0, "."); import evdr_amd, bench as B
from evdr_amd.corpus import PageCorpus
dev = torch.device("cuda:0"); pages = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
P = B.gen_pages(0, pages, dev); corpus = PageCorpus.from_tensor(P, None)
Qall, _ = B.make_queries(1024, pages, P, 0, pages, dev, 1)
NQS = tuple(int(x) for x in sys.argv[2].split(',')) if len(sys.argv) > 2 else (1, 2, 4, 8, 12, 16, 24, 32, 40, 64, 128, 256, 1024)
for nq in NQS:
    Q = Qall[:nq].contiguous(); out = torch.empty((nq, pages), dtype=torch.float32, device=dev)
    corpus.score(Q, None, out=out); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): corpus.score(Q, None, out=out)
    b.record(); torch.cuda.synchronize(); ms = a.elapsed_time(b) / 5
    gb = pages * 263680 / 1e9
    print(f"nq={nq:5d}  {ms:8.3f} ms  corpus stream {gb/ms*1e3/1e3:6.2f} TB/s  {nq*pages*B.FLOP_PER_PAIR/ms/1e9:8.1f} TFLOP/s  {nq/ms*1e3*pages/1e5:9.1f} q/s@100k", flush=True)
