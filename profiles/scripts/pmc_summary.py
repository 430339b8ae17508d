#!/usr/bin/env python3
"""Sum the rocprofv3 counter_collection csv files of a pmc pass directory per kernel (last dispatch of every kernel name
= the profiled step) and print one table.  Usage: pmc_summary.py gpurun_out/pmc_<tag>_a gpurun_out/pmc_<tag>_b ..."""
import csv, glob, sys, collections, re
tab = collections.OrderedDict()
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        # keep the LAST dispatch of every kernel (steady state)
        last = {}
        for r in rows:
            k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
            last.setdefault(k, {})
            disp = int(r["Dispatch_Id"])
            last[k].setdefault(disp, {})[r["Counter_Name"]] = float(r["Counter_Value"])
        for k, byd in last.items():
            d_ = byd[max(byd)]
            tab.setdefault(k, {}).update(d_)
names = sorted({c for v in tab.values() for c in v})
print("kernel\t" + "\t".join(names))
for k, v in tab.items():
    print(k + "\t" + "\t".join("%.4g" % v.get(c, float("nan")) for c in names))
