import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv") + glob.glob(sys.argv[1] + "/*kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(r["Name"][:80].ljust(80), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
