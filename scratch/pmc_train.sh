#!/bin/bash
# Round 5: PMC passes over the fused training step (BASELINE.json configs[4]) for its three kernels -- teacher
# forward, student forward + argmax, update (maxsim_bwd_kernel<..,true>) -- each pass its own run, rocprofv3 with --kernel-trace only,
# the program directly after `--`.  Aggregated on the box into one JSON (per mode and kernel: mean counter values per launch).
#   modes: fused (teacher stream non-temporal = product), fused_nt_off (variant 34: default cache policy), fused_cached (no teacher launch)
# usage: bash scratch/pmc_train.sh r05
set -o pipefail
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_train_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "MALL|DRAM|EA0_RDREQ|EA0_WRREQ|TCC_HIT|TCC_MISS|TCC_REQ" | cut -c1-160 | sort -u | head -60 > $OUT/counters_available.txt
REGEX="maxsim_fwd16s_kernel|maxsim_bwd_kernel|infonce_row_kernel"
run() { mode=$1; name=$2; shift 2; extra=""; only=$mode
        if [ "$mode" = "fused_nt_off" ]; then only=fused; extra="--fwd-variant 34"; fi
        rm -rf /tmp/pmct_${mode}_$name
        timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "$REGEX" --pmc "$@" --output-format csv -d /tmp/pmct_${mode}_$name -o $name -- \
            python3 $R/bench_train.py --steps 20 --warmup 10 --only $only --no-cpu-baseline --no-roofline $extra > $OUT/${mode}_$name.json 2> $OUT/${mode}_$name.err
        echo "$mode $name exit=$?"; }
run fused fetch FETCH_SIZE &&
run fused write WRITE_SIZE &&
run fused sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE &&
run fused sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU &&
run fused l2 TCC_HIT_sum TCC_MISS_sum &&
run fused_nt_off fetch FETCH_SIZE &&
run fused_nt_off write WRITE_SIZE &&
run fused_nt_off l2 TCC_HIT_sum TCC_MISS_sum &&
run fused_cached fetch FETCH_SIZE &&
run fused_cached write WRITE_SIZE &&
run fused_cached sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE &&
run fused_cached l2 TCC_HIT_sum TCC_MISS_sum
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections, os
out = sys.argv[1]
def role(name):
    if "maxsim_bwd_kernel" in name: return "update"
    if "infonce" in name: return "loss"
    if "maxsim_fwd16s_kernel<2, 2, true" in name or "maxsim_fwd16s_kernel<1, 2, true" in name: return "student"
    if "maxsim_fwd16s_kernel" in name: return "teacher"
    return None
summary = collections.defaultdict(lambda: collections.defaultdict(dict))
names = collections.defaultdict(dict)
for d in sorted(glob.glob("/tmp/pmct_*")):
    mode_pass = os.path.basename(d)[len("pmct_"):]
    mode = mode_pass.rsplit("_", 1)[0]
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            ro = role(r["Kernel_Name"])
            if ro is None: continue
            names[mode][ro] = r["Kernel_Name"]
            agg[(ro, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (ro, c), v in agg.items():
            v = v[len(v) // 3:]                          # drop the warm-up third (same shapes; clocks and caches settled)
            summary[mode][ro][c] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            ro = role(r["Kernel_Name"])
            if ro: agg[ro].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for ro, v in agg.items():
            v = v[len(v) // 3:]
            summary[mode][ro].setdefault("_kernel_ns_under_pmc", {})[mode_pass] = {"launches": len(v), "mean": sum(v) / len(v)}
json.dump({"kernels": names, "counters": summary}, open(out + "/pmc_train_raw.json", "w"), indent=1)
print(json.dumps({m: {r: sorted(c) for r, c in v.items()} for m, v in summary.items()}, indent=1)[:3000])
PY
