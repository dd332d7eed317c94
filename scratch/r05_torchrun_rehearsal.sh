#!/bin/bash
# round 5 (PageCorpus.topk reworked, TORCH_NCCL_ASYNC_ERROR_HANDLING=0): the driver's launch form (torch.distributed.run) on the 1-GPU box: 1 rank with the RCCL data group (--dist-at-one), 2 ranks over gloo
mkdir -p gpurun_out/r05
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 1 --dist-at-one --pages 3000 --queries 64 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-other-regimes > gpurun_out/r05/torchrun1_nccl.json 2> gpurun_out/r05/torchrun1_nccl.err; echo "torchrun 1 rank nccl rc=$?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29632 bench.py --gpus 2 --backend gloo --pages 3000 --queries 64 --steps 2 --warmup 1 --no-cpu-baseline --no-other-regimes > gpurun_out/r05/torchrun2_gloo.json 2> gpurun_out/r05/torchrun2_gloo.err; echo "torchrun 2 ranks gloo rc=$?"
EVDR_BENCH_ALLOW_SHARED_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29633 bench.py --gpus 2 --pages 3000 --queries 64 --steps 2 --warmup 1 --no-cpu-baseline --no-other-regimes > gpurun_out/r05/torchrun2_nccl_refused.json 2> gpurun_out/r05/torchrun2_nccl_refused.err; echo "torchrun 2 ranks nccl-refused rc=$?"
python - <<PY
import json
for f in ("torchrun1_nccl", "torchrun2_gloo", "torchrun2_nccl_refused"):
    try:
        r = json.loads([l for l in open(f"gpurun_out/r05/{f}.json") if l.startswith("{")][-1])
        d = r["dist"]; print(f, d["world_size"], d["ranks_seen"], d["backend"], d["launcher"], (d["backend_fallback_reason"] or "")[:60], r["ndcg_at_5"], r["phases"].get("max_over_ranks"))
    except Exception as e:
        print(f, "no line:", e); print(open(f"gpurun_out/r05/{f}.err").read()[-1500:])
PY
