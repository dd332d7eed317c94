#!/bin/bash
# Round-3 experiment session on one MI355X box (VERDICT r2 item 5 + first GPU run of the shard-loading / overlap changes).
# usage: bash scratch/r03_experiments.sh     -> gpurun_out/r03_exp_*.{log,txt,json}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_driver.py tests/test_gpu_edge_semantics.py tests/test_gpu_parity.py -q -x -m gpu > $O/r03_exp_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r03_exp_tests.log
# (b) student forward beside the teacher forward on a second stream: A/B of the step, twice, alternating
for rep in 1 2; do
  timeout -k 10 200 python3 bench_train.py --only fused,fused_overlap,fused_nosync,fused_overlap_nosync --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null \
    | python3 -c "import json,sys; d=json.load(sys.stdin); print({k: round(v['ms_per_step'],4) for k,v in d['results'].items()})" >> $O/r03_exp_overlap.txt
done
echo "overlap:"; cat $O/r03_exp_overlap.txt
# (a) LDS-fed MFMA loops of both shapes at equal LDS bytes per FLOP, with socket power and clock sampled beside
( for i in $(seq 1 60); do echo "t=$(date +%s.%N | cut -c1-13) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Current Socket Graphics Package Power|sclk clock level' | tr '\n' ' ' | sed -e 's/GPU\[0\]\s*: //g' -e 's/=\+//g')"; sleep 0.4; done ) > $O/r03_exp_shapes_power.txt 2>&1 &
SMI=$!
timeout -k 10 120 $R/scratch/probe/mfma_lds_shapes > $O/r03_exp_shapes.txt 2>&1; echo "probe rc=$?"
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
cat $O/r03_exp_shapes.txt
# multi-rank rehearsals of bench.py on the one GPU: gloo exchange (phases + ranks_seen in the line), then the nccl request that RCCL
# must refuse (two ranks on one device) -> the gloo fallback path
cd /tmp
timeout -k 10 200 python3 $R/bench.py --gpus 2 --backend gloo --pages 8000 --queries 256 --steps 3 --warmup 1 > $O/r03_exp_gloo2.json 2> $O/r03_exp_gloo2.err; echo "gloo2 rc=$?"
EVDR_BENCH_ALLOW_SHARED_GPU=1 timeout -k 10 200 python3 $R/bench.py --gpus 2 --backend nccl --pages 8000 --queries 256 --steps 3 --warmup 1 > $O/r03_exp_nccl_fallback.json 2> $O/r03_exp_nccl_fallback.err; echo "nccl-fallback rc=$?"
head -c 1500 $O/r03_exp_nccl_fallback.json; tail -5 $O/r03_exp_nccl_fallback.err
