"""Host side of the reference's call pattern with the score-row cache on: cProfile over 300 steps (each with its float(loss) wait),
plus the step time with and without that wait -- where the 0.1 ms between the kernels' sum and the step goes."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd, bench_train as BT
from evdr_amd.criterion import infonce_distillation_loss
from evdr_amd.evaluator import retrieval as R
from evdr_amd.utils.preprocess_data import l2_normalize
from evdr_amd.utils.utils import set_optimizer
dev = torch.device("cuda:0")
inp = BT.make_inputs(500, 32, dev)
B, Pt, pmt, pms, Qall, qmall = inp["B"], inp["Pt"], inp["pmt"], inp["pms"], inp["Qall"], inp["qmall"]
param = torch.nn.Parameter(inp["Pbar0"].clone()); opt = set_optimizer("adamw", param, 1e-3, 1e-2)
R.enable_score_cache(2 << 30)
order = torch.arange(64 * B, device=dev)


def step(i, sync=True):
    lo = (i % 64) * B
    idx = order[lo:lo + B]
    Qb, qmb = Qall.index_select(0, idx), qmall.index_select(0, idx)
    Psb = l2_normalize(param * pms.unsqueeze(-1))
    with torch.no_grad():
        sc_t = R.score_multi_vector_masked(Qb, Pt, qmb, pmt, 64)
    sc_s = R.score_multi_vector_masked(Qb, Psb, qmb, pms, 64)
    loss = infonce_distillation_loss(sc_s, sc_t, temperature=0.1)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return float(loss.item()) if sync else loss


for i in range(128): step(i)
torch.cuda.synchronize()
for sync in (True, False):
    t0 = time.perf_counter()
    for i in range(300): step(i, sync)
    torch.cuda.synchronize()
    print(f"sync={sync}: {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms per step", flush=True)
pr = cProfile.Profile(); pr.enable()
for i in range(300): step(i, False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(32)

# ---- host time per block of the step (perf_counter between the blocks, no device sync inside the loop)
import collections
acc = collections.OrderedDict()
def lap(name, t):
    now = time.perf_counter(); acc[name] = acc.get(name, 0.0) + now - t; return now
torch.cuda.synchronize()
for i in range(300):
    t = time.perf_counter()
    lo = (i % 64) * B; idx = order[lo:lo + B]
    Qb, qmb = Qall.index_select(0, idx), qmall.index_select(0, idx); t = lap("index_select x2", t)
    Psb = l2_normalize(param * pms.unsqueeze(-1)); t = lap("mul + l2_normalize", t)
    with torch.no_grad():
        sc_t = R.score_multi_vector_masked(Qb, Pt, qmb, pmt, 64)
    t = lap("teacher score (cached)", t)
    sc_s = R.score_multi_vector_masked(Qb, Psb, qmb, pms, 64); t = lap("student score", t)
    loss = infonce_distillation_loss(sc_s, sc_t, temperature=0.1); t = lap("loss", t)
    opt.zero_grad(set_to_none=True); t = lap("zero_grad", t)
    loss.backward(); t = lap("backward", t)
    opt.step(); t = lap("opt.step", t)
torch.cuda.synchronize()
print("host microseconds per step, by block (no sync in the loop):")
for k, v in acc.items(): print(f"  {k:26s} {v / 300 * 1e6:7.1f}")
print(f"  {'total':26s} {sum(acc.values()) / 300 * 1e6:7.1f}")
