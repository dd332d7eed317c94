"""Print the kernels of the last few training steps from a rocprofv3 kernel trace: start offset, duration, gap to the previous kernel."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at the query split (absmax over the query batch) -- find teacher forwards and cut between them
names = [r["Kernel_Name"] for r in rows]
bw = [i for i, n in enumerate(names) if "maxsim_bwd_kernel" in n]
steps = [i for j, i in enumerate(bw) if j > 0 and i - bw[j - 1] > 3]      # backward launches that close a whole step
a, b = steps[-3], steps[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
print(f"{'start us':>9s} {'dur us':>8s} {'gap us':>7s}  kernel")
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {r['Kernel_Name'][:110]}")
    prev_end = e
span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3 / 2
print(f"step period (backward to backward): {span:.1f} us")
