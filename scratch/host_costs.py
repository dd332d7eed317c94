"""Host-side enqueue cost (us) of each piece of the fused training step: time.perf_counter around the call, device idle before
and synchronised after (so nothing queues up behind a busy GPU)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
from evdr_amd import driver, ops
from evdr_amd.utils.preprocess_data import l2_normalize
dev = torch.device("cuda:0"); torch.manual_seed(0)
N, B, Lt, Ls = 500, 32, 1030, 206
Pt = l2_normalize(torch.randn(N, Lt, 128, device=dev)); pmt = torch.ones(N, Lt, dtype=torch.bool, device=dev)
pms = torch.ones(N, Ls, dtype=torch.bool, device=dev)
teacher = driver.TeacherScorer(Pt, pmt); student = driver.FusedStudent(torch.randn(N, Ls, 128, device=dev), pms, 1e-3, 1e-2)
Qall = l2_normalize(torch.randn(64 * B, 32, 128, device=dev)); qmall = torch.ones(64 * B, 32, dtype=torch.bool, device=dev)
def host(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); tot = 0.0
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); tot += time.perf_counter() - t0; torch.cuda.synchronize()
    return tot / n * 1e6
idx = torch.arange(B)
Qb, qmb = Qall[idx], qmall[idx]
qpl = ops.split_f32(Qb)
sc_t = teacher.scores(Qb, qmb, qplanes=qpl); sc_s, arg = student.scores(Qb, qmb, qpl)
loss, ds = ops.infonce_distill(sc_s, sc_t, 0.1, want_grad=True)
rows = [("harness: Qall[idx], qmall[idx]", lambda: (Qall[idx], qmall[idx])),
        ("Qb.to(dev).float(), qmb.to(dev)", lambda: (Qb.to(dev, non_blocking=True).float(), qmb.to(dev, non_blocking=True))),
        ("ops.split_f32(Qb)", lambda: ops.split_f32(Qb)),
        ("teacher.scores", lambda: teacher.scores(Qb, qmb, qplanes=qpl)),
        ("student.scores", lambda: student.scores(Qb, qmb, qpl)),
        ("ops.infonce_distill", lambda: ops.infonce_distill(sc_s, sc_t, 0.1, want_grad=True)),
        ("student.apply", lambda: student.apply(ds, Qb, qmb, arg)),
        ("loss.item()", lambda: loss.item()),
        ("whole fused_train_one_step(sync=False)", lambda: driver.fused_train_one_step(Qb, qmb, teacher, student, 0.1, sync=False))]
for name, fn in rows:
    print(f"{name:42s} {host(fn):7.1f} us", flush=True)
