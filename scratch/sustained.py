"""Back-to-back launches of one forward shape for several seconds: per-launch time over time (DVFS ramp / steady state).
usage: python scratch/sustained.py <nq> <pages> [seconds]"""
import sys, time, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd, bench as B
from evdr_amd.corpus import PageCorpus
nq, pages = int(sys.argv[1]), int(sys.argv[2]); secs = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
dev = torch.device("cuda:0")
P = B.gen_pages(0, pages, dev); corpus = PageCorpus.from_tensor(P, None)
Q, _ = B.make_queries(max(nq, 32), pages, P, 0, pages, dev, 1); Q = Q[:nq].contiguous()
out = torch.empty((nq, pages), dtype=torch.float32, device=dev)
corpus.score(Q, None, out=out); torch.cuda.synchronize()
t_end = time.time() + secs; batch = 50; rows = []
while time.time() < t_end:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(batch): corpus.score(Q, None, out=out)
    b.record(); torch.cuda.synchronize(); rows.append(a.elapsed_time(b) / batch)
n = len(rows)
print(f"nq={nq} pages={pages}: {n} batches of {batch}; ms/launch first {rows[0]:.4f}  median {sorted(rows)[n//2]:.4f}  last {rows[-1]:.4f}  min {min(rows):.4f}  "
      f"-> {pages*263680/sorted(rows)[n//2]/1e6:.0f} GB/s, {nq*pages*B.FLOP_PER_PAIR/sorted(rows)[n//2]/1e9:.0f} TF", flush=True)
