import sys, torch
sys.path.insert(0, "."); import evdr_amd
from evdr_amd import ops
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(78)
for scale in (1.0, 3e4, 2e-7):
    x = (torch.randn(37, 19, 128, generator=gen) * scale).to(dev)
    planes, amax = ops.split_f32(x)
    torch.cuda.synchronize()
    print(scale, hex(int(amax.item()) & 0xFFFFFFFF), hex(int(x.abs().max().view(torch.int32).item())), float(x.abs().max()), float(amax.view(torch.float32).item()))
