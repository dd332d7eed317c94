#!/bin/bash
# Socket power and shader clock while the training step's teacher forward runs back to back (see power_probe.sh).
R=$GRAFT_REPO_ROOT
python3 $R/scratch/sustained_train.py 6 > /tmp/sus.log 2>&1 &
PID=$!
sleep 3.5
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Current Socket Graphics Package Power|sclk clock level|mclk clock level" | tr '\n' ' ' | sed -e 's/GPU\[0\]\s*: //g' -e 's/=\+//g'
  echo
  sleep 0.3
done
wait $PID
grep -v amdgpu.ids /tmp/sus.log | tail -1
