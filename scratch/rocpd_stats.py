"""Kernel statistics (same columns as rocprofv3's kernel_stats.csv) from a rocprofv3 rocpd database.
usage: python scratch/rocpd_stats.py results.db out.csv"""
import csv, sqlite3, statistics, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, end - start from kernels").fetchall()
agg = {}
for n, d in rows:
    agg.setdefault(n, []).append(d)
tot = sum(sum(v) for v in agg.values())
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([n, len(v), sum(v), round(sum(v) / len(v), 3), round(100.0 * sum(v) / tot, 2), min(v), max(v),
                    round(statistics.pstdev(v), 3)])
