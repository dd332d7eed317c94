"""Top-k kernel time: rows of MaxSim-like scores (clustered values) and of wide-range values."""
import sys, torch
sys.path.insert(0, "."); import evdr_amd
from evdr_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
for nq, n in ((1024, 100000), (1024, 12500), (1, 100000), (500, 6847)):
    for name, s in (("clustered", 5.5 + 0.3 * torch.randn(nq, n, device=dev)), ("wide", torch.randn(nq, n, device=dev) * torch.exp(3 * torch.randn(nq, n, device=dev)))):
        ts, ti = ops.topk(s, 100); torch.cuda.synchronize()
        ws, wi = torch.topk(s, 100, dim=1)
        assert torch.equal(ts, ws), "top-k values differ from torch.topk"
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): ops.topk(s, 100)
        b.record(); torch.cuda.synchronize()
        print(f"nq={nq:5d} n={n:6d} {name:9s}: {a.elapsed_time(b)/5*1e3:9.1f} us  ({nq*n*4/ (a.elapsed_time(b)/5*1e-3)/1e12:5.2f} TB/s of one row read)", flush=True)
