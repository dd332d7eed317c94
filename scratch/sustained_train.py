"""Back-to-back launches of the training step's teacher forward (32 x 500 x 1030, fp32 as fp16 hi/lo planes) for a few seconds:
per-launch time (for scratch/power_probe.sh-style sampling of socket power and clock).  usage: python scratch/sustained_train.py [seconds]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
from evdr_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0); secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
def unit(*s): return torch.nn.functional.normalize(torch.randn(*s, device=dev), dim=-1)
B, N, Lt = 32, 500, 1030
Q, Pt = unit(B, 32, 128), unit(N, Lt, 128)
qp, qa = ops.split_f32(Q); tp, ta = ops.split_f32(Pt); tm, pf = ops.pack_pmask(None, N, Lt, dev)
out = torch.empty(B, N, device=dev)
fn = lambda: ops.maxsim_forward_prepared(qp, qa, tp, ta, None, tm, pf, out=out)
fn(); torch.cuda.synchronize()
t_end = time.time() + secs; rows = []
while time.time() < t_end:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): fn()
    b.record(); torch.cuda.synchronize(); rows.append(a.elapsed_time(b) / 200)
med = sorted(rows)[len(rows) // 2]
print(f"teacher forward 32 x 500 x 1030 fp16 hi/lo: {len(rows)} batches of 200; us/launch first {rows[0]*1e3:.1f} median {med*1e3:.1f} min {min(rows)*1e3:.1f} "
      f"-> {2.0*B*N*32*Lt*128*3/med/1e9:.0f} TFLOP/s of plane products", flush=True)
