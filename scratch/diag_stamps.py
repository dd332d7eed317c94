"""In-kernel stamp shares of the staged MaxSim kernel (diagnostic instantiation; shares only, not run time)."""
import os, sys, torch
import _hooks as H
EXP = H.use_experiment_build()          # libevdr_exp.so: the only build that carries the stamped instances
sys.path.insert(0, "."); import evdr_amd, bench as B
from evdr_amd.corpus import PageCorpus
dev = torch.device("cuda:0"); pages = 20000
P = B.gen_pages(0, pages, dev); corpus = PageCorpus.from_tensor(P, None)
Q, _ = B.make_queries(1024, pages, P, 0, pages, dev, 1)
out = torch.empty((1024, pages), dtype=torch.float32, device=dev)
dbg = torch.zeros((4096, 8, 8), dtype=torch.int64, device=dev)
EXP.evdr_experiment_set_dbg_buffer(dbg.data_ptr()); H.set_variant(sys.argv[1] if len(sys.argv) > 1 else "50")
for _ in range(3): corpus.score(Q, None, out=out)
torch.cuda.synchronize()
d = dbg.cpu().double(); d = d[d[:, :, 0] > 0]            # waves that reported
tot = d[:, 0]
names = ["total", "prologue", "barrier wait", "refill at stage top", "fast block", "generic / tail tile", "page finish", "control before barrier"]
NST = 61 * 4   # pages per workgroup x stages per page at 20 000 pages (informative only)
print(f"waves reporting: {len(d)}, mean total cycles {tot.mean():.0f}")
for i, n in enumerate(names[1:], 1):
    print(f"{n:22s} {100*(d[:, i]/tot).mean():6.2f} %   ({d[:, i].mean()/NST:8.0f} cycles per stage)")
print(f"{'unaccounted':22s} {100*(1 - (d[:,1:8].sum(1)/tot)).mean():6.2f} %")
