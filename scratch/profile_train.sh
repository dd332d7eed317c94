#!/bin/bash
# The training half of a round's artefacts on one box (scratch/profile_round.sh without the retrieval bench and the PMC passes).
set -o pipefail
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench_train.py --steps 60 > $O/${TAG}_bench_train.json 2> $O/${TAG}_bench_train.err; echo "train rc=$?"
rm -rf /tmp/prof_t; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o train -- python3 $R/bench_train.py --steps 30 --only fused --no-cpu-baseline --no-roofline > $O/${TAG}_prof_train.json 2> $O/${TAG}_prof_train.err; echo "prof train rc=$?"
cp $(ls /tmp/prof_t/*/*kernel_stats.csv /tmp/prof_t/*kernel_stats.csv 2>/dev/null | head -1) $O/${TAG}_train_fused_kernel_stats.csv
