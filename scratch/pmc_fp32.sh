#!/bin/bash
# MFMA-busy fraction and effective clock of the fp32 (fp16 hi/lo) forward kernel at 500 x 6847 x 1030
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_fp32; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --kernel-include-regex "maxsim_fwd16s" --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmc_fp32 -o sq -- python3 $R/scratch/fp32_workload.py > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_fp32/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "maxsim_fwd16s" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for f in glob.glob("/tmp/pmc_fp32/*kernel_trace.csv") for r in csv.DictReader(open(f)) if "maxsim_fwd16s" in r["Kernel_Name"]]
t = sum(d) / len(d) * 1e-9
m = {k: sum(v) / len(v) for k, v in agg.items()}
cyc = m["GRBM_GUI_ACTIVE"] / 8
s = {"workload": "500 queries x 6847 pages x 1030 patches, fp32 inputs as fp16 hi/lo planes (maxsim_fwd16s_kernel<2,2,false,4,2,false,true,2,true>)",
     "kernel_ms": t * 1e3, "effective_clock_ghz": cyc / t / 1e9, "mfma_busy_frac": m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc,
     "plane_product_tflops": 500 * 6847 * 2 * 32 * 1030 * 128 * 3 / t / 1e12, "lds_bank_conflict_cycles": m.get("SQ_LDS_BANK_CONFLICT"), "launches": len(d)}
json.dump(s, open(sys.argv[1] + "/pmc_fp32_summary.json", "w"), indent=1)
print(json.dumps(s, indent=1))
PY
