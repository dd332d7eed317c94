#!/bin/bash
# kernel statistics of the reference's CALL PATTERN (autograd + set_optimizer's AdamW) and of the fused step, timed steps only
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -o cp -- python3 $R/bench_train.py --steps 30 --only call_pattern --no-cpu-baseline --no-roofline > $O/prof_callpattern.json 2> $O/prof_callpattern.err; echo "rc=$?"
cp $(ls /tmp/prof_c/*/*kernel_stats.csv /tmp/prof_c/*kernel_stats.csv 2>/dev/null | head -1) $O/callpattern_kernel_stats.csv
python3 $R/scratch/trace_exclusive.py --last 30 $(ls /tmp/prof_c/*/*kernel_trace.csv /tmp/prof_c/*kernel_trace.csv 2>/dev/null | head -1) > $O/callpattern_trace_last30.json
rm -rf /tmp/prof_t; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o train -- python3 $R/bench_train.py --steps 30 --only fused --no-cpu-baseline --no-roofline > $O/prof_fused.json 2> $O/prof_fused.err; echo "rc=$?"
python3 $R/scratch/trace_exclusive.py --last 30 $(ls /tmp/prof_t/*/*kernel_trace.csv /tmp/prof_t/*kernel_trace.csv 2>/dev/null | head -1) "maxsim_fwd16s_kernel<2, 2, false" "maxsim_fwd16s_kernel<2, 2, true" maxsim_bwd_kernel infonce_row_kernel split_small_kernel vectorized_gather copyBuffer > $O/fused_trace_last30.json
cat $O/prof_callpattern.json | python3 -c "import json,sys; r=json.load(sys.stdin); print(r['results'])"
cat $O/prof_fused.json | python3 -c "import json,sys; r=json.load(sys.stdin); print(r['results'])"
cat $O/fused_trace_last30.json
python3 - <<PY
import json
r=json.load(open("$O/callpattern_trace_last30.json"))
tot=0
for k,v in sorted(r["kernels"].items(), key=lambda kv:-kv[1]["plain_avg_us"]*kv[1]["calls"]):
    per_step=v["plain_avg_us"]*v["calls"]/30; tot+=per_step
    print(f"{k[-60:]:60s} calls/step {v['calls']/30:4.1f}  avg {v['plain_avg_us']:8.1f} us  per step {per_step:7.1f} us")
print("sum per step", round(tot,1), "us")
PY
