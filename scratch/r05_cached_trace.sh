#!/bin/bash
# in-step kernel trace of the fused step WITH cached teacher scores (the steady state of a run: every pseudo-query is scored by the frozen teacher once)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_fc; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fc -o fc -- python3 $R/bench_train.py --steps 60 --only fused_cached --no-cpu-baseline --no-roofline > $O/r05_prof_train_cached_line.json 2> $O/r05_prof_train_cached.err; echo "rc=$?"
python3 $R/scratch/trace_exclusive.py --last 60 $(ls /tmp/prof_fc/*/*kernel_trace.csv /tmp/prof_fc/*kernel_trace.csv 2>/dev/null | head -1) "maxsim_fwd16s_kernel<2, 2, false" "maxsim_fwd16s_kernel<2, 2, true" maxsim_bwd_kernel infonce_row_kernel split_small_kernel split_segments_kernel vectorized_gather copyBuffer > $O/r05_train_cached_trace_exclusive.json
cat $O/r05_train_cached_trace_exclusive.json; python3 -c "import json; print(json.load(open('$O/r05_prof_train_cached_line.json'))['results'])"
