"""Short view of a rocprofv3 kernel_stats.csv: python scratch/kstats.py file.csv [n]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in rows[:n]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"\(.*", "", name)[:78]
    print(f"{name:78s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.2f} us  min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f}")
