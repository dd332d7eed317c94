#!/bin/bash
# Socket power and shader clock (rocm-smi, sampled every ~0.3 s) while one forward shape runs back to back for a few seconds.
# usage (GPU box): bash scratch/power_probe.sh "<nq> <pages>" ["<nq> <pages>" ...]
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  set -- $cfg
  python3 $R/scratch/sustained.py $1 $2 6 > /tmp/sus.log 2>&1 &
  PID=$!
  sleep 3.0   # past torch import + corpus build for the small shapes; samples taken while the loop runs
  for i in 1 2 3 4 5 6; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|mclk clock level" | tr '\n' ' ' | sed -e 's/GPU\[0\]\s*: //g' -e 's/=\+//g'
    echo
    sleep 0.3
  done
  wait $PID
  echo "cfg nq=$1 pages=$2: $(grep -v amdgpu.ids /tmp/sus.log | tail -1)"
done
