"""A/B of the fused step's float(loss) hand-over: the loss kernel storing straight into the pinned host word (direct_loss_store,
round 4) against the device scalar + copy launch of rounds 2-3; same process, interleaved blocks of steps; the losses must agree bit for bit."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
import bench_train as BT
from evdr_amd import driver
dev = torch.device("cuda:0")
inp = BT.make_inputs(500, 32, dev)
B, Qall, qmall = inp["B"], inp["Qall"], inp["qmall"]
teacher = driver.TeacherScorer(inp["Pt"], inp["pmt"])
S = {d: driver.FusedStudent(inp["Pbar0"].clone(), inp["pms"], lr=1e-3, weight_decay=1e-2) for d in (True, False)}
for d, s in S.items(): s.direct_loss_store = d
tot = {True: 0.0, False: 0.0}; losses = {True: [], False: []}
for rnd in range(6):
    for d, s in S.items():
        for i in range(20): driver.fused_train_one_step(Qall[(i % 64) * B:(i % 64 + 1) * B], qmall[(i % 64) * B:(i % 64 + 1) * B], teacher, s, 0.1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(60):
            losses[d].append(driver.fused_train_one_step(Qall[(i % 64) * B:(i % 64 + 1) * B], qmall[(i % 64) * B:(i % 64 + 1) * B], teacher, s, 0.1))
        torch.cuda.synchronize(); tot[d] += (time.perf_counter() - t0) / 60 * 1e3
print(f"fused step with float(loss) every step: direct store {tot[True] / 6:.4f} ms   copy launch {tot[False] / 6:.4f} ms   "
      f"losses bit-equal: {losses[True] == losses[False]} ({len(losses[True])} steps)")
