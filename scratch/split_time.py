"""Time of ops.split_f32 (absmax + fp16 hi/lo split) on page-sized fp32 tensors.  usage: python scratch/split_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evdr_amd  # noqa: F401
from evdr_amd import ops
dev = "cuda:0"
for shape in ((500, 206, 128), (500, 1030, 128), (6847, 1030, 128)):
    x = torch.randn(shape, device=dev)
    for _ in range(5):
        ops.split_f32(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        ops.split_f32(x)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    mb = x.numel() * 4 / 1e6
    print(f"split_f32 {shape}: {us:8.1f} us  ({mb:.0f} MB fp32, read twice + written once as planes: {3 * mb * 1e6 / (us * 1e-6) / 1e12:.2f} TB/s)")
