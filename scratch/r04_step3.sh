#!/bin/bash
# round-4 GPU step 3: the touched tests, the N>1 rehearsals of bench.py on one GPU (gloo; nccl refused -> agreed fallback), then the
# round's profile artefacts (PMC passes, bench line, rocprofv3 kernel stats, train bench + in-step profile, nq sweep)
set -o pipefail
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout -k 10 400 python -m pytest tests/test_gpu_driver.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/gputests_c.log 2>&1
echo "pytest rc=$?" >> $O/gputests_c.log; tail -4 $O/gputests_c.log
grep -q "pytest rc=0" $O/gputests_c.log || exit 1
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --pages 4000 --queries 64 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"
EVDR_BENCH_ALLOW_SHARED_GPU=1 timeout -k 10 600 python bench.py --gpus 2 --backend nccl --pages 4000 --queries 64 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_nccl_refused.json 2> $O/bench_nccl_refused.err; echo "nccl-refused rc=$?"
python - <<PY
import json
for f in ("bench_gloo2", "bench_nccl_refused"):
    try:
        r = json.loads([l for l in open(f"gpurun_out/r04/{f}.json") if l.startswith("{")][-1])
        print(f, r["dist"], r["phases"].get("max_over_ranks"), r["ndcg_at_5"])
    except Exception as e:
        print(f, "no line:", e)
PY
bash scratch/profile_round.sh r04
