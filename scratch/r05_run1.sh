#!/bin/bash
# round 5, GPU call 1: the whole GPU suite on the new source, then the sentinel control with the staged-ring WAR build
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout -k 10 1000 python -m pytest tests -m gpu -q -s 2>&1 | tail -40 > gpurun_out/r05_gputests_1.txt
rc=$?
tail -5 gpurun_out/r05_gputests_1.txt
echo "pytest rc=$rc"
timeout -k 10 300 python scratch/sentinel_control.py > gpurun_out/r05_sentinel_control.txt 2>&1
cat gpurun_out/r05_sentinel_control.txt
