import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
n = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
for r in rows[:24]:
    print("%8.2f us/step %6.2f calls/step avg %7.1f us  %s" % (float(r["TotalDurationNs"]) / 1e3 / n, int(r["Calls"]) / n, float(r["AverageNs"]) / 1e3, r["Name"][:90]))
print("total us/step", sum(float(r["TotalDurationNs"]) for r in rows) / 1e3 / n)
