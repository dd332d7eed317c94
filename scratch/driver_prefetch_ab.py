"""Three datasets (the same synthetic dumps under three names; 500 pages x ~1030 teacher patches, 3200 pseudo-queries) through the driver:
ONE run over all three (the loader thread reads and pads dataset k+1 while dataset k trains) against three single-dataset runs in a row
(every dataset read in front of its training).  usage: python scratch/driver_prefetch_ab.py [steps per dataset]"""
import io, json, os, sys, tempfile, time, contextlib
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import evdr_amd  # noqa: F401
from evdr_amd import driver
from test_gpu_driver import write_synthetic_dataset
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
tmp = Path(tempfile.mkdtemp()); write_synthetic_dataset(tmp, n_pages=500, lt=1030, mf=4, n_train=3200)
m = json.loads((tmp / "map.json").read_text())
names = ["synth", "synth_b", "synth_c"]
for n in names[1:]:
    m[n] = dict(m["synth"])
(tmp / "map3.json").write_text(json.dumps(m))
def run(datasets, tag):
    argv = ["--datasets"] + datasets + ["--mapping_json", str(tmp / "map3.json"), "--query_root", str(tmp), "--teacher_root", str(tmp),
            "--init_root", str(tmp), "--mfs", "4", "--out_root", str(tmp / tag), "--name", "run", "--max_steps", str(steps), "--eval_every", "1000",
            "--print_every", "1000", "--q_batch", "32", "--fused_step", "--cache_teacher_scores"]
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        driver.main(argv)
    return time.time() - t0
run(["synth"], "warm")
t_sep = [run([n], "sep_" + n) for n in names]
t_one = run(names, "one")
print(f"three single-dataset runs: {' + '.join(f'{t:.2f}' for t in t_sep)} = {sum(t_sep):.2f} s;  one run over the three datasets: {t_one:.2f} s "
      f"({steps} fused steps with cached teacher scores + {steps // 1000 + 1} evaluations per dataset)")
