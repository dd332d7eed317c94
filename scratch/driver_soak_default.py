"""`python -m evdr_amd.driver` with the REFERENCE's flags alone (round 5: the fused step and the teacher score cache are the default) at
the BASELINE.json configs[4] shape -- 500 pages, teacher ~1030 patches, student ~206 (mf 5), q_batch 32, fp32 -- on a synthetic npz
dataset in the reference's schema: ms per step from the log's own `time_sec` (per 100-step line; the first epoch also fills the teacher
cache), next to the same run with --no_fused_step --no_cache_teacher_scores (the reference's call pattern) and the logged losses of the
two compared.  usage: python scratch/driver_soak_default.py [steps]"""
import json, os, subprocess, sys, tempfile, time
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_driver import write_synthetic_dataset
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
tmp = Path(tempfile.mkdtemp())
t0 = time.time()
write_synthetic_dataset(tmp, n_pages=500, lt=1030, mf=5, n_train=4096)
print(f"dataset written in {time.time() - t0:.1f} s: 500 pages x <= 1030 patches, 4096 pseudo-queries x 12 tokens, student = 5-patch block means", flush=True)
env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
runs = {}
for tag, extra, n in (("default (reference flags only)", [], steps), ("--no_fused_step --no_cache_teacher_scores", ["--no_fused_step", "--no_cache_teacher_scores"], steps // 2)):
    out = tmp / ("r_" + str(len(runs)))
    cmd = [sys.executable, "-m", "evdr_amd.driver", "--datasets", "synth", "--mapping_json", str(tmp / "map.json"), "--query_root", str(tmp),
           "--teacher_root", str(tmp), "--init_root", str(tmp), "--mfs", "4", "--out_root", str(out), "--name", "run", "--max_steps", str(n),
           "--eval_every", str(n), "--print_every", "100", "--q_batch", "32"] + extra
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    print(f"--- {tag}: wall {wall:.1f} s; the driver said:")
    for ln in r.stdout.splitlines():
        if ln.startswith("[fast paths]") or ln.startswith("[save]") or ln.startswith("[done]"):
            print("    " + ln)
    lines = (out / "run" / "mf4" / "synth" / "train.log").read_text().splitlines()
    recs = [json.loads(ln[ln.index("{"):]) for ln in lines if "{" in ln]
    tr = [r_ for r_ in recs if "train/loss" in r_]
    runs[tag] = tr
    prev_t, prev_s = 0.0, 0
    for r_ in tr:
        dt = (r_["time_sec"] - prev_t) * 1e3 / (r_["step"] - prev_s)
        print(f"    step {r_['step']:5d}  loss {r_['train/loss']:.6f}  avg {r_['train/avg_loss']:.6f}  {dt:.3f} ms per step over the last {r_['step'] - prev_s} steps")
        prev_t, prev_s = r_["time_sec"], r_["step"]
    summ = [r_ for r_ in recs if "summary/best_ndcg5" in r_][-1]
    print(f"    summary: {json.dumps(summ)}", flush=True)
a, b = runs["default (reference flags only)"], runs["--no_fused_step --no_cache_teacher_scores"]
common = min(len(a), len(b))
dl = max(abs(x["train/loss"] - y["train/loss"]) / max(abs(y["train/loss"]), 1e-9) for x, y in zip(a[:common], b[:common]))
print(f"max relative difference of the logged losses, default vs the reference's call pattern, over the first {common} log lines: {dl:.2e}")
assert dl < 5e-3
print("ok")
