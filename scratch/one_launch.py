"""A few forward launches of one shape (for rocprofv3): python scratch/one_launch.py <nq> <pages> [reps] [mask]"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd, bench as B
from evdr_amd.corpus import PageCorpus
nq, pages = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
P = B.gen_pages(0, pages, dev); corpus = PageCorpus.from_tensor(P, None)
Q, _ = B.make_queries(max(nq, 32), pages, P, 0, pages, dev, 1); Q = Q[:nq].contiguous()
out = torch.empty((nq, pages), dtype=torch.float32, device=dev)
for _ in range(reps): corpus.score(Q, None, out=out)
torch.cuda.synchronize()
