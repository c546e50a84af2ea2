"""Turn a rocprofv3 rocpd (.db) kernel trace into the --stats style per-kernel summary CSV."""
import csv
import sqlite3
import statistics
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = c.execute(f"select {name_col}, start, end from kernels").fetchall()
by = {}
for n, s, e in rows:
    by.setdefault(n, []).append(e - s)
tot = sum(sum(v) for v in by.values())
with open(out, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([n, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v),
                    statistics.stdev(v) if len(v) > 1 else 0.0])
