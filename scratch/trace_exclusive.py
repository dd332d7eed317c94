"""Per-kernel time of a rocprofv3 --kernel-trace CSV as the steps see it.

rocprofv3's per-dispatch interval [Start, End] opens when the dispatch packet is taken up, which for back-to-back launches on one
in-order queue is BEFORE the kernel in front has drained: consecutive intervals overlap, their plain averages (the --stats CSV)
add up to more than the wall time they sit in.  This script walks the dispatches of one process in start order and reports, per
kernel name, (a) the plain average (= the --stats figure), (b) the EXCLUSIVE average: End - max(Start, End of the dispatch in
front), i.e. the time by which the kernel extends the queue's busy period, and the overlap statistics.  The exclusive figures of
the kernels of a step add up to the step's device time.
--last N: only the last N dispatches of each kernel (the TIMED steps of bench_train.py; its warm-up steps run on a colder, slower chip
and would otherwise sit in the averages).
usage: python scratch/trace_exclusive.py [--last N] <kernel_trace.csv> [<name substring> ...]   -> JSON on stdout"""
import csv
import json
import sys

last = 0
if sys.argv[1] == "--last":
    last = int(sys.argv[2])
    del sys.argv[1:3]
rows = list(csv.DictReader(open(sys.argv[1])))
subs = sys.argv[2:]
d = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
out = {}
prev_end = None
n_overlap = 0
per = {}
for s, e, name in d:
    ov = max(0, prev_end - s) if prev_end is not None else 0
    n_overlap += ov > 0
    ex = e - max(s, prev_end) if prev_end is not None else e - s
    prev_end = max(prev_end or e, e)
    key = next((x for x in subs if x in name), None) if subs else name.split("(")[0][-70:]
    if key is None:
        continue
    per.setdefault(key, []).append((e - s, max(ex, 0), ov))
for key, lst in per.items():
    if last:
        lst = lst[-last:]
    out[key] = {"calls": len(lst), "plain_ns": sum(x[0] for x in lst), "exclusive_ns": sum(x[1] for x in lst),
                "overlap_with_predecessor_ns": sum(x[2] for x in lst)}
for o in out.values():
    for k in ("plain_ns", "exclusive_ns", "overlap_with_predecessor_ns"):
        o[k.replace("_ns", "_avg_us")] = round(o.pop(k) / o["calls"] / 1e3, 2)
print(json.dumps({"last_calls_per_kernel": last or None, "dispatches": len(d), "dispatches_overlapping_their_predecessor": n_overlap, "kernels": out}, indent=1))
