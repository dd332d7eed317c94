"""Query-batch split (32 x 32 x 128 fp32 -> fp16 hi/lo planes): microseconds per call, events around 200 calls."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
from evdr_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
for shape in ((32, 32, 128), (8, 32, 128), (64, 32, 128), (1, 32, 128)):
    Q = torch.nn.functional.normalize(torch.randn(*shape, device=dev), dim=-1)
    for _ in range(10): ops.split_f32(Q)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(200): ops.split_f32(Q)
    b.record(); torch.cuda.synchronize()
    print(shape, f"{a.elapsed_time(b) / 200 * 1e3:.1f} us per split (host-bound if > kernel time)")
