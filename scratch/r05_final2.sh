#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
bash scratch/r05_torchrun_rehearsal.sh > gpurun_out/r05_torchrun_rehearsal.txt 2>&1; cat gpurun_out/r05_torchrun_rehearsal.txt | tail -8
bash scratch/profile_round.sh r05
