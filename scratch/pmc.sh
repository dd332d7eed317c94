#!/bin/bash
# PMC passes for the bench workload, MaxSim kernel only; raw CSVs are aggregated on the box into one small JSON.
# (separate passes: TCC slots do not fit FETCH_SIZE + WRITE_SIZE together -- MI355X_MICROARCH.md "rocprofv3 PMC slots")
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_${1:-r01}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-other-regimes --no-extras ${BENCH_EXTRA:-}"
run() { name=$1; shift; timeout -k 10 280 rocprofv3 --kernel-trace --kernel-include-regex "maxsim_fwd" --pmc "$@" --output-format csv -d /tmp/pmc_$name -o $name -- python3 $R/bench.py $ARGS > $OUT/$name.json 2> $OUT/$name.err; echo "$name exit=$?"; }
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE &&
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
summary = {}
for f in sorted(glob.glob("/tmp/pmc_*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "maxsim_fwd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        summary[k] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
for f in sorted(glob.glob("/tmp/pmc_*/*kernel_trace.csv")):
    d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if "maxsim_fwd" in r["Kernel_Name"]]
    if d: summary.setdefault("_kernel_ns_under_pmc", {})[f.split("/")[-1]] = {"launches": len(d), "mean": sum(d)/len(d)}
json.dump(summary, open(out + "/pmc_summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
