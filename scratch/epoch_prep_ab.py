"""Fused step with the epoch's batches prepared once per epoch (driver.EpochBatches: "fused", "fused_nosync") against the
per-step gather + split ("fused_stepprep", "fused_stepprep_nosync"), interleaved rounds in one process; bench_train.time_mode
of each (the epoch preparation lies inside the timed region).  usage: epoch_prep_ab.py [rounds] [steps]"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd  # noqa
import bench_train as BT
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
inp = BT.make_inputs(500, 32, torch.device("cuda:0"))
kinds = ["fused_stepprep", "fused", "fused_stepprep_nosync", "fused_nosync", "fused_cached_nosync"]
res = {k: [] for k in kinds}
for r in range(rounds):
    for k in kinds:
        res[k].append(BT.time_mode(inp, k, steps, 20)["ms_per_step"])
for k in kinds:
    print(f"{k:24s} median {statistics.median(res[k]):.4f} ms  ({' '.join(f'{v:.4f}' for v in res[k])})", flush=True)
