"""cProfile of the host side of the fused training step (300 steps, no per-step sync)."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
from evdr_amd import driver
from evdr_amd.utils.preprocess_data import l2_normalize
dev = torch.device("cuda:0"); torch.manual_seed(0)
N, B, Lt, Ls = 500, 32, 1030, 206
Pt = l2_normalize(torch.randn(N, Lt, 128, device=dev)); pmt = torch.ones(N, Lt, dtype=torch.bool, device=dev)
pms = torch.ones(N, Ls, dtype=torch.bool, device=dev)
teacher = driver.TeacherScorer(Pt, pmt); student = driver.FusedStudent(torch.randn(N, Ls, 128, device=dev), pms, 1e-3, 1e-2)
Qb = l2_normalize(torch.randn(B, 32, 128, device=dev)); qmb = torch.ones(B, 32, dtype=torch.bool, device=dev)
for _ in range(20): driver.fused_train_one_step(Qb, qmb, teacher, student, 0.1, sync=False)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): driver.fused_train_one_step(Qb, qmb, teacher, student, 0.1, sync=False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
