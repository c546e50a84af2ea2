"""Per-kernel mean of a rocprofv3 --pmc counter from its counter_collection CSV."""
import csv
import sys

path = sys.argv[1]
by = {}
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].split("(")[0]
    by.setdefault((name, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (name, ctr), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"{ctr:12s} {name[:70]:70s} dispatches {len(v):5d}  mean {sum(v) / len(v):16.1f}  total {sum(v):18.1f}")
