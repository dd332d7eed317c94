"""Interleaved A/B timing of forward-kernel variants in ONE process (cdna_hip_programming.md rule 24).
usage: python scratch/kbench.py [pages] [variants comma list] [rounds]"""
import _hooks as H
import os, sys, time, torch
sys.path.insert(0, "."); import evdr_amd
from evdr_amd.corpus import PageCorpus
sys.path.insert(0, "."); import bench as B
pages = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2").split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
zero = len(sys.argv) > 4 and sys.argv[4] == 'zero'
dev = torch.device("cuda:0")
P = B.gen_pages(0, pages, dev)
Q0 = None
Q, _ = B.make_queries(1024, pages, P, 0, pages, dev, 1)
if zero:
    P.zero_(); Q.zero_(); print('ALL-ZERO DATA (clock experiment)')
corpus = PageCorpus.from_tensor(P, None)
out = torch.empty((1024, pages), dtype=torch.float32, device=dev)
ref = None
res = {v: [] for v in variants}
for rnd in range(rounds + 1):
    for v in variants:
        H.set_variant(v)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); corpus.score(Q, None, out=out); b.record(); torch.cuda.synchronize()
        if rnd == 0:
            if ref is None: ref = out.clone()
            else:
                d = (out - ref).abs().max().item()
                print(f"variant {v}: max|diff vs variant {variants[0]}| = {d:.3e}", flush=True)
        else:
            res[v].append(a.elapsed_time(b))
flop = 1024 * pages * B.FLOP_PER_PAIR
for v in variants:
    ts = sorted(res[v]); med = ts[len(ts)//2]
    print(f"variant {v}: median {med:8.2f} ms  min {ts[0]:8.2f}  -> {flop/med/1e9:7.1f} TFLOP/s (median), {flop/ts[0]/1e9:7.1f} (best)", flush=True)
