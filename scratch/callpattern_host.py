"""Host-side cost of the reference's call pattern (bench_train.py 'call_pattern'): wall per step, and the top of a cProfile of 300 steps."""
import cProfile, io, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
import bench_train as BT
dev = torch.device("cuda:0")
inp = BT.make_inputs(500, 32, dev)
for k in ("call_pattern", "cached", "resident", "call_pattern"):
    r = BT.time_mode(inp, k, 100, 30)
    print(k, round(r["ms_per_step"], 4), flush=True)
pr = cProfile.Profile()
pr.enable()
BT.time_mode(inp, "call_pattern", 300, 10)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
