#!/usr/bin/env python3
"""Per-kernel average of rocprofv3 --pmc counters (csv counter_collection file)."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0][-48:]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[k][row["Counter_Name"]] += 1
for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
    if k.startswith("void at::") or "rocclr" in k:
        continue
    print(k, {c: (agg[k][c] / cnt[k][c], cnt[k][c]) for c in agg[k]})
