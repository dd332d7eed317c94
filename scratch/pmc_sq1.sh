#!/bin/bash
# one SQ/GRBM pass for a kernel variant: MFMA-busy fraction and effective clock
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_$1; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --kernel-include-regex "maxsim_fwd" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_$1 -o sq1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --pages ${PAGES:-100000} > $OUT/sq1.json 2> $OUT/sq1.err
python3 - "$OUT" "$1" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(f"/tmp/pmc_{tag}/*counter_collection.csv")[0])):
    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(glob.glob(f"/tmp/pmc_{tag}/*kernel_trace.csv")[0]))]
m = {k: sum(v)/len(v) for k, v in agg.items()}
t = sum(d)/len(d)*1e-9; cyc = m["GRBM_GUI_ACTIVE"]/8
res = {"tag": tag, "kernel_ms": t*1e3, "clock_ghz": cyc/t/1e9, "mfma_busy": m["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/cyc,
       "wait_any": m["SQ_WAIT_ANY"]/m["SQ_WAVE_CYCLES"], "wait_inst": m["SQ_WAIT_INST_ANY"]/m["SQ_WAVE_CYCLES"], "raw": m}
json.dump(res, open(out + "/sq1_summary.json", "w"), indent=1); print({k: v for k, v in res.items() if k != "raw"})
PY
