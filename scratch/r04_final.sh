#!/bin/bash
# round-4 closing GPU step: the whole GPU suite once (it includes the pass through the sentinel build), then the round's profile artefacts
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputests_final.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04/gputests_final.log
tail -4 gpurun_out/r04/gputests_final.log
grep -q "pytest rc=0" gpurun_out/r04/gputests_final.log || exit 1
bash scratch/profile_round.sh r04
