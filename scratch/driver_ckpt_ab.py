"""The driver end to end at the docvqa_test_subsampled shape (500 pages x ~1030 teacher patches, mf 4 -> ~258 student patches, 3200
pseudo-queries, evaluation every epoch of 100 steps) with the best-checkpoint files written inline (--sync_checkpoints: what the
reference does) against the background writer (default).  Same seeds, same files; prints the wall time of the two runs and the
number of checkpoints written.  usage: python scratch/driver_ckpt_ab.py [steps]"""
import io, json, os, sys, tempfile, time, contextlib
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import evdr_amd  # noqa: F401
from evdr_amd import driver
from test_gpu_driver import write_synthetic_dataset
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
if "--improve" in sys.argv:
    # the first epochs of a real run: EVERY evaluation improves both best metrics (this synthetic student peaks at its first evaluation,
    # which would leave the writers nothing to do); the decision rule is replaced for both legs alike
    driver.update_best = lambda best, metrics, step, kind: ({"step": step, "Recall@1": float(metrics["Recall"]["Recall@1"]),
                                                            "NDCG@5": float(metrics["NDCG"]["NDCG@5"])}, True)
tmp = Path(tempfile.mkdtemp()); t0 = time.time()
write_synthetic_dataset(tmp, n_pages=500, lt=1030, mf=4, n_train=3200)
print(f"dataset written in {time.time() - t0:.1f} s", flush=True)
res = {}
for tag, extra in (("warm-up (not reported)", ["--max_steps", "20"]), ("inline (--sync_checkpoints)", ["--sync_checkpoints"]), ("background writer (default)", [])):
    out = tmp / ("r_" + tag.split()[0])
    argv = ["--datasets", "synth", "--mapping_json", str(tmp / "map.json"), "--query_root", str(tmp), "--teacher_root", str(tmp),
            "--init_root", str(tmp), "--mfs", "4", "--out_root", str(out), "--name", "run", "--max_steps", str(steps), "--eval_every", "100",
            "--print_every", "100", "--q_batch", "32", "--fused_step", "--cache_teacher_scores"] + extra
    buf = io.StringIO(); t0 = time.time()
    with contextlib.redirect_stdout(buf):
        driver.main(argv)
    dt = time.time() - t0
    saves = buf.getvalue().count("[save]")
    log = (out / "run" / "mf4" / "synth" / "train.log").read_text().splitlines()
    summ = [json.loads(ln[ln.index("{"):]) for ln in log if "summary/best_ndcg5" in ln][-1]
    res[tag] = (dt, saves, summ["summary/best_ndcg5"])
    if not tag.startswith("warm"):
        print(f"{tag:32s} {dt:7.2f} s wall for {steps} steps + {steps // 100 + 1} evaluations, {saves} checkpoint files written, best {summ['summary/best_ndcg5']}", flush=True)
a, b = res["inline (--sync_checkpoints)"], res["background writer (default)"]
assert a[2] == b[2]
print(f"checkpoint files: inline {a[1]}, background {b[1]} (waiting snapshots that a newer one superseded are not written)")
za = np.load(tmp / "r_inline" / "run" / "mf4" / "synth" / "best_ndcg5.npz", allow_pickle=True)
zb = np.load(tmp / "r_background" / "run" / "mf4" / "synth" / "best_ndcg5.npz", allow_pickle=True)
assert all(np.array_equal(x, y) for x, y in zip(za["documents"], zb["documents"])) and za["meta"].item()["step"] == zb["meta"].item()["step"]
print(f"same best checkpoint (step {za['meta'].item()['step']}); wall time {a[0]:.2f} -> {b[0]:.2f} s")
