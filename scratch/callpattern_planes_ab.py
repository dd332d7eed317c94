"""Reference call pattern (bench_train "call_pattern": the drop-in l2_normalize / score_multi_vector_masked / loss / optimizer used
exactly like mainv2_iter_distill_infonce.py:269-292) with the planes hand-over from l2_normalize to the scorer (default) against
the same with the hand-over disabled (ops._DERIVED_MAX = 0: the scorer runs absmax + split over Psb again; l2_normalize still
writes the planes, so the 'off' leg is a few us slower than the code before the change).  Interleaved rounds, one process."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd  # noqa
import bench_train as BT
from evdr_amd import ops
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
inp = BT.make_inputs(500, 32, torch.device("cuda:0"))
res = {"on": [], "off": []}
for r in range(rounds):
    for tag, cap in (("off", 0), ("on", 8)):
        ops._DERIVED_MAX = cap
        ops._DERIVED.clear()
        res[tag].append(BT.time_mode(inp, "call_pattern", 60, 20)["ms_per_step"])
ops._DERIVED_MAX = 8
for tag in ("off", "on"):
    print(f"hand-over {tag:3s}: median {statistics.median(res[tag]):.4f} ms  ({' '.join(f'{v:.4f}' for v in res[tag])})", flush=True)
