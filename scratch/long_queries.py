"""Queries padded to 50 tokens, most of them shorter than 32 valid tokens: cost of the second 32-token slice."""
import sys, torch
sys.path.insert(0, "."); import evdr_amd, bench as B
from evdr_amd import ops
dev = torch.device("cuda:0"); pages = 20000
P = B.gen_pages(0, pages, dev)
g = torch.Generator(device=dev).manual_seed(3)
nq = 512
for frac_long in (0.0, 0.1, 0.5, 1.0):
    lens = torch.where(torch.rand(nq, generator=g, device=dev) < frac_long, torch.randint(33, 51, (nq,), generator=g, device=dev),
                       torch.randint(12, 33, (nq,), generator=g, device=dev))
    qm = torch.arange(50, device=dev)[None, :] < lens[:, None]
    Q = torch.nn.functional.normalize(torch.randn(nq, 50, 128, generator=g, device=dev), dim=-1).bfloat16()
    out, _ = ops.maxsim_forward(Q, P, qm, None); torch.cuda.synchronize()
    ref, _ = ops.maxsim_forward(Q[:, :32].contiguous(), P, qm[:, :32].contiguous(), None)
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.maxsim_forward(Q, P, qm, None); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.maxsim_forward(Q[:, :32].contiguous(), P, qm[:, :32].contiguous(), None); b.record(); torch.cuda.synchronize()
    short = (lens <= 32)
    same = bool(torch.equal(out[short], ref[short]))
    print(f"{frac_long*100:5.0f} % of queries longer than 32 tokens: {min(ts):8.2f} ms   (first slice alone {a.elapsed_time(b):8.2f} ms)   short queries unchanged: {same}", flush=True)
