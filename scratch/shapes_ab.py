"""Forward time over the shapes of BASELINE.json's configs; EVDR_LIB_AB=<path> loads another build of the library."""
import os, sys, torch
sys.path.insert(0, "."); import evdr_amd
from evdr_amd import _lib
if os.environ.get("EVDR_LIB_AB"): _lib.LIB_PATH = os.environ["EVDR_LIB_AB"]
import bench as B
from evdr_amd import ops
dev = torch.device("cuda:0")
shapes = [(32, 40000, 1030), (500, 500, 1030), (500, 6847, 1030), (32, 500, 1030), (32, 500, 206), (1024, 12500, 1030), (8, 40000, 1030), (64, 6847, 1030)]
Pbig = B.gen_pages(0, 40000, dev)
for nq, np_, lp in shapes:
    P = Pbig[:np_, :lp].contiguous()
    Q = torch.nn.functional.normalize(torch.randn(nq, 32, 128, device=dev), dim=-1).to(torch.bfloat16)
    for _ in range(2): out, _ = ops.maxsim_forward(Q, P, None, None, want_argmax=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    a.record()
    for _ in range(reps): out, _ = ops.maxsim_forward(Q, P, None, None, want_argmax=False)
    b.record(); torch.cuda.synchronize(); ms = a.elapsed_time(b) / reps
    print(f"nq={nq:5d} np={np_:6d} lp={lp:5d}  {ms*1e3:10.1f} us  {nq*np_*2*32*lp*128/ms/1e9:8.1f} TFLOP/s  chk {float(out.double().sum()):.6f}", flush=True)
