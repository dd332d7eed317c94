"""The driver end to end on the synthetic npz dataset of tests/test_gpu_driver.py for several hundred steps: autograd path, fused path
(--fused_step: planes kept by the update kernel, losses read in batches) and fused + cached teacher scores; compares the logged
loss trajectories and final metrics.  usage: python scratch/driver_soak.py [steps]"""
import json, os, sys, tempfile
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import evdr_amd  # noqa: F401
from evdr_amd import driver
from test_gpu_driver import write_synthetic_dataset
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
tmp = Path(tempfile.mkdtemp()); write_synthetic_dataset(tmp, n_pages=96, n_train=768)

runs = {}
for tag, extra in (("autograd", []), ("fused", ["--fused_step"]), ("fused_cached", ["--fused_step", "--cache_teacher_scores"])):
    out = tmp / ("r_" + tag)
    driver.main(["--datasets", "synth", "--mapping_json", str(tmp / "map.json"), "--query_root", str(tmp), "--teacher_root", str(tmp),
                 "--init_root", str(tmp), "--mfs", "4", "--out_root", str(out), "--name", "run", "--max_steps", str(steps),
                 "--eval_every", str(steps // 4), "--print_every", "25", "--q_batch", "32"] + extra)
    lines = (out / "run" / "mf4" / "synth" / "train.log").read_text().splitlines()
    recs = [json.loads(ln[ln.index("{"):]) for ln in lines if "{" in ln]
    runs[tag] = ([r["train/loss"] for r in recs if "train/loss" in r], [r["train/avg_loss"] for r in recs if "train/avg_loss" in r],
                 [r for r in recs if "summary/best_ndcg5" in r][-1])
ref = runs["autograd"]
for tag, (loss, avg, summ) in runs.items():
    dl = max(abs(a - b) / max(abs(b), 1e-9) for a, b in zip(loss, ref[0]))
    print(f"{tag:13s} {len(loss)} log lines  last loss {loss[-1]:.6f}  last avg {avg[-1]:.6f}  max rel diff of logged losses vs autograd {dl:.2e}  "
          f"best nDCG@5 {summ['summary/best_ndcg5']['NDCG@5']:.5f} Recall@1 {summ['summary/best_recall']['Recall@1']:.5f}", flush=True)
    assert all(np.isfinite(loss)) and dl < 5e-3
print("ok")
