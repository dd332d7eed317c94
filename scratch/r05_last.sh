#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05_gputests_last_tail.txt; echo "pytest rc=$?"; tail -3 gpurun_out/r05_gputests_last_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
timeout -k 10 400 python bench.py 2>/dev/null | tail -1 > gpurun_out/r05_bench_last.json; python -c "
import json; d = json.load(open('gpurun_out/r05_bench_last.json')); print('bench:', round(d['value']/1e6, 2), 'M pairs/s', round(d['ms_per_step'], 2), 'ms; frac', round(d['roofline']['frac'], 4), '; train', {k: round(v['ms_per_step'], 4) for k, v in d['train_step']['results'].items()})"
