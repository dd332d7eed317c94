"""A few launches of the fp32 (fp16 hi/lo) forward at the configs[2] shape, for profiling."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
from evdr_amd import ops
from evdr_amd.corpus import PageCorpus
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
P = torch.nn.functional.normalize(torch.randn((6847, 1030, 128), generator=g, device=dev), dim=-1)
Q = torch.nn.functional.normalize(torch.randn((500, 32, 128), generator=g, device=dev), dim=-1)
c = PageCorpus.from_tensor(P); del P
out = torch.empty((500, 6847), dtype=torch.float32, device=dev)
for _ in range(4): c.score(Q, None, out=out)
torch.cuda.synchronize()
