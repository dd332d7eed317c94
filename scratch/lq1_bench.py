"""Single-token ("virtual") queries: time of score_multi_vector_masked-style forward at Lq = 1."""
import sys, torch
sys.path.insert(0, "."); import evdr_amd
from evdr_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
def unit(*s): return torch.nn.functional.normalize(torch.randn(*s, device=dev), dim=-1)
P = unit(500, 1030, 128); Ps = unit(500, 206, 128)
for nq in (32, 128, 512):
    Q = unit(nq, 1, 128)
    for name, X in (("teacher 1030", P), ("student 206", Ps)):
        ops.maxsim_forward(Q, X, None, None, want_argmax=True); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): ops.maxsim_forward(Q, X, None, None, want_argmax=True)
        b.record(); torch.cuda.synchronize()
        print(f"nq={nq:4d} Lq=1 {name}: {a.elapsed_time(b)/10*1e3:8.1f} us (fp32 inputs, argmax, incl. split of P)", flush=True)
