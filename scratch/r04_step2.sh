#!/bin/bash
# round-4 GPU step: targeted tests after the backward rewrite, then the training bench with in-step kernel times
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_edge_semantics.py tests/test_gpu_driver.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sharded_train.py tests/test_v3_patterns.py -m gpu -x -q > gpurun_out/r04/gputests_b.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04/gputests_b.log
tail -15 gpurun_out/r04/gputests_b.log
grep -q "pytest rc=0" gpurun_out/r04/gputests_b.log || exit 1
timeout -k 10 200 python bench_train.py --steps 60 --only fused,fused_nosync,fused_cached --no-cpu-baseline > gpurun_out/r04/bench_train_a.json 2> gpurun_out/r04/bench_train_a.err || { tail -5 gpurun_out/r04/bench_train_a.err; exit 1; }
python - <<PY
import json
r=json.load(open("gpurun_out/r04/bench_train_a.json"))
print({k:round(v["ms_per_step"],4) for k,v in r["results"].items()})
for e in r["roofline"]:
    print(e["role"][:30], round(e["kernel_ms"]*1e3,1), "us alone", round(e["kernel_ms_in_step"]*1e3,1), "us in step", round(e["frac"],3), round(e["frac_in_step"],3))
PY
