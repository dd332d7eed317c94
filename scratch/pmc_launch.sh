#!/bin/bash
# SQ counter passes for ONE forward shape: scratch/pmc_launch.sh <tag> <nq> <pages>
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "maxsim_fwd" --pmc "$@" --output-format csv -d /tmp/pmcl_$name -o $name -- python3 $R/scratch/one_launch.py $NQ $PAGES > $OUT/$name.log 2>&1; echo "$name exit=$?"; }
NQ=$2; PAGES=$3
rm -rf /tmp/pmcl_*
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE &&
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU &&
run sq3 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVES
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
summary = {}
for f in sorted(glob.glob("/tmp/pmcl_*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "maxsim_fwd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        summary[k] = {"launches": len(v), "mean": sum(v) / len(v)}
for f in sorted(glob.glob("/tmp/pmcl_*/*kernel_trace.csv")):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if "maxsim_fwd" in r["Kernel_Name"]]
    if d: summary.setdefault("_kernel_ns_under_pmc", {})[f.split("/")[-1]] = {"launches": len(d), "mean": sum(d) / len(d)}
json.dump(summary, open(out + "/pmc_summary.json", "w"), indent=1)
for k, v in summary.items(): print(k, v)
PY
