#!/bin/bash
# One fused training step as the GPU saw it: kernel start offsets, durations and the idle gaps between them.
# usage (on the GPU box): bash scratch/train_timeline.sh [mode]
set -o pipefail
MODE=${1:-fused_nosync}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o tl -- python3 $R/bench_train.py --steps 12 --warmup 3 --only $MODE --no-cpu-baseline > /tmp/tl.json 2> /tmp/tl.err || { tail -5 /tmp/tl.err; exit 1; }
python3 $R/scratch/train_timeline.py $(ls /tmp/prof_tl/*/*kernel_trace.csv /tmp/prof_tl/*kernel_trace.csv 2>/dev/null | head -1)
