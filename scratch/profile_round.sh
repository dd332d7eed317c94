#!/bin/bash
# Round artefacts on one box: PMC passes first (they make profiles/hbm_traffic.json, which the bench line replays with the file's
# hash), then the bench line, rocprofv3 kernel stats of the same command, train bench + its kernel stats, nq sweep.
# usage: bash scratch/profile_round.sh r02      (afterwards, locally: python scratch/pmc_post.py r02 -> the same derived files)
set -o pipefail
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
bash $R/scratch/pmc.sh $TAG > $O/${TAG}_pmc.log 2>&1; echo "pmc rc=$?"
(cd $R && python3 scratch/pmc_post.py $TAG > $O/${TAG}_pmc_post.log 2>&1); echo "pmc_post rc=$?"
cd /tmp
python3 $R/bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err; echo "bench rc=$?"
rm -rf /tmp/prof_b; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof_bench.err; echo "prof bench rc=$?"
cp $(ls /tmp/prof_b/*/*kernel_stats.csv /tmp/prof_b/*kernel_stats.csv 2>/dev/null | head -1) $O/${TAG}_bench_kernel_stats.csv
python3 $R/bench_train.py --steps 60 > $O/${TAG}_bench_train.json 2> $O/${TAG}_bench_train.err; echo "train rc=$?"
rm -rf /tmp/prof_t; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o train -- python3 $R/bench_train.py --steps 30 --only fused --no-cpu-baseline --no-roofline > $O/${TAG}_prof_train.json 2> $O/${TAG}_prof_train.err; echo "prof train rc=$?"
cp $(ls /tmp/prof_t/*/*kernel_stats.csv /tmp/prof_t/*kernel_stats.csv 2>/dev/null | head -1) $O/${TAG}_train_fused_kernel_stats.csv
# the same trace per dispatch: rocprofv3's intervals of back-to-back launches overlap; exclusive (queue-extending) time per kernel
python3 $R/scratch/trace_exclusive.py --last 30 $(ls /tmp/prof_t/*/*kernel_trace.csv /tmp/prof_t/*kernel_trace.csv 2>/dev/null | head -1) "maxsim_fwd16s_kernel<2, 2, false" "maxsim_fwd16s_kernel<2, 2, true" maxsim_bwd_kernel infonce_row_kernel split_small_kernel split_segments_kernel vectorized_gather copyBuffer > $O/${TAG}_train_fused_trace_exclusive.json; echo "trace_exclusive rc=$?"
cd $R && python3 scratch/small_nq.py 40000 2>&1 | grep -v amdgpu.ids > $O/${TAG}_nq_sweep.txt; echo "sweep rc=$?"
cat $O/${TAG}_bench_n1.json | head -c 600; echo; head -4 $O/${TAG}_bench_kernel_stats.csv | cut -c1-200; cat $O/${TAG}_nq_sweep.txt
