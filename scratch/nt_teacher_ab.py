"""Fused training step (configs[4]) with the teacher forward's corpus stream on the default cache policy (variant 34) against
the non-temporal policy (variant 0 = the dispatch rule, which picks it for this shape; 33 = forced; 34 = default policy forced), interleaved in one process.  The question: the frozen teacher corpus (264 MB as fp16
hi/lo planes) is streamed once per step through the 256-MiB Infinity Cache and evicts the student state (parameters, both
moments, planes: 211 MB) that the update kernel and the student forward re-read a few hundred microseconds later.  Reports
the step time and the in-step time of each kernel; losses of the two variants must be bit-equal (same arithmetic)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd  # noqa
import bench_train as BT
from evdr_amd import _lib as L, driver, ops

dev = torch.device("cuda:0")
lib = L.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
inp = BT.make_inputs(N, 32, dev)
B, Qall, qmall = inp["B"], inp["Qall"], inp["qmall"]
variants = tuple(int(v) for v in sys.argv[4].split(",")) if len(sys.argv) > 4 else (34, 0)   # 34 = default policy forced, 0 = the dispatch rule (nt for this shape), 33 = nt forced
state = {}
for v in variants:
    state[v] = (driver.TeacherScorer(inp["Pt"], inp["pmt"]), driver.FusedStudent(inp["Pbar0"].clone(), inp["pms"], lr=1e-3, weight_decay=1e-2))

rec = {v: {"teacher": [], "student": [], "update": []} for v in variants}
cur = [None]
orig_f, orig_u = ops.maxsim_forward_prepared, ops.maxsim_backward_adamw


def bracket(role, fn, *a, **k):
    if cur[0] is None:
        return fn(*a, **k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = fn(*a, **k); e1.record()
    rec[cur[0]][role].append((e0, e1))
    return r


ops.maxsim_forward_prepared = lambda *a, **k: bracket("student" if k.get("want_argmax") else "teacher", orig_f, *a, **k)
ops.maxsim_backward_adamw = lambda *a, **k: bracket("update", orig_u, *a, **k)

tot = {v: [] for v in variants}
losses = {v: [] for v in variants}
names = {}
it = {v: 0 for v in variants}
for rnd in range(rounds):
    for v in variants:
        teacher, student = state[v]
        lib.evdr_debug_set_fwd_variant(v)
        for phase in ("warm", "timed"):
            n = 15 if phase == "warm" else steps
            cur[0] = v if phase == "timed" else None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n):
                lo = (it[v] % 64) * B; it[v] += 1
                last = driver.fused_train_one_step(Qall[lo:lo + B], qmall[lo:lo + B], teacher, student, 0.1, sync=False)
            torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t0) / n
        cur[0] = None
        tot[v].append(dt)
        losses[v].append(float(last))
lib.evdr_debug_set_fwd_variant(0)
for v in variants:
    ks = {r: 1e3 * sum(a.elapsed_time(b) for a, b in ev) / len(ev) for r, ev in rec[v].items()}
    print(f"variant {v:2d}: step {sum(tot[v]) / len(tot[v]):.4f} ms (rounds {' '.join(f'{t:.4f}' for t in tot[v])})  in-step us: teacher {ks['teacher']:.1f} "
          f"student {ks['student']:.1f} update {ks['update']:.1f}", flush=True)
print("losses equal:", all(losses[v] == losses[variants[0]] for v in variants), [losses[v][-1] for v in variants])
