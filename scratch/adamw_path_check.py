"""Which path does utils.StreamAdamW.step() take for the training step's parameter, and what does a step cost on the host?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
from evdr_amd import ops
from evdr_amd.utils.utils import StreamAdamW
dev = torch.device("cuda:0")
p = torch.nn.Parameter(torch.randn(500, 206, 128, device=dev))
opt = StreamAdamW([p], lr=1e-3, weight_decay=1e-2)
calls = [0]
orig = ops.adamw_step
ops.adamw_step = lambda *a, **k: (calls.__setitem__(0, calls[0] + 1), orig(*a, **k))[1]
p.grad = torch.randn_like(p)
print("eligible:", opt._eligible(p), "grad ptr%16", p.grad.data_ptr() % 16, "contig", p.grad.is_contiguous())
for _ in range(3): opt.step()
print("kernel-path calls in 3 steps:", calls[0], {k: (v.device, v.dtype) if torch.is_tensor(v) else v for k, v in opt.state[p].items()})
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): opt.step()
torch.cuda.synchronize(); print(f"StreamAdamW.step: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per step (device + host)")
t0 = time.perf_counter()
for _ in range(200): opt._eligible(p)
print(f"_eligible alone: {(time.perf_counter() - t0) / 200 * 1e6:.2f} us")
o2 = torch.optim.AdamW([p], lr=1e-3, weight_decay=1e-2)
for _ in range(3): o2.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): o2.step()
torch.cuda.synchronize(); print(f"torch.optim.AdamW.step: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per step")
