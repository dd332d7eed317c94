"""Fused training step, per-step float(loss) (sync) against loss left on the device (nosync), alternating in one process; and how
long the host sits in wait_loss().  usage: python scratch/sync_modes.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
from evdr_amd import driver
from evdr_amd.utils.preprocess_data import l2_normalize
dev = torch.device("cuda:0"); torch.manual_seed(0)
N, B, Lt, Ls = 500, 32, 1030, 206
Pt = l2_normalize(torch.randn(N, Lt, 128, device=dev)); pmt = torch.ones(N, Lt, dtype=torch.bool, device=dev)
pms = torch.ones(N, Ls, dtype=torch.bool, device=dev)
teacher = driver.TeacherScorer(Pt, pmt); student = driver.FusedStudent(torch.randn(N, Ls, 128, device=dev), pms, 1e-3, 1e-2)
Qall = l2_normalize(torch.randn(64 * B, 32, 128, device=dev)); qmall = torch.ones(64 * B, 32, dtype=torch.bool, device=dev)
order = torch.arange(64 * B, device=dev)
def step(i, sync):
    idx = order[(i % 64) * B:(i % 64 + 1) * B]
    return driver.fused_train_one_step(Qall.index_select(0, idx), qmall.index_select(0, idx), teacher, student, 0.1, sync=sync)
def run(sync, n=100):
    for i in range(5): step(i, sync)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): step(i, sync)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rnd in range(3):
    print(f"round {rnd}: sync {run(True):.4f} ms   nosync {run(False):.4f} ms", flush=True)
# where the synchronous step's host time goes
waits, enq = [], []
orig = student.wait_loss
def timed_wait():
    t0 = time.perf_counter(); v = orig(); waits.append(time.perf_counter() - t0); return v
student.wait_loss = timed_wait
torch.cuda.synchronize()
for i in range(100):
    t0 = time.perf_counter(); step(i, True); enq.append(time.perf_counter() - t0)
torch.cuda.synchronize()
import statistics as S
print(f"sync step: host total {S.mean(enq)*1e6:.0f} us of which waiting for the loss {S.mean(waits)*1e6:.0f} us (enqueue {S.mean(enq)*1e6 - S.mean(waits)*1e6:.0f} us)")
