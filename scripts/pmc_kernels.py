#!/usr/bin/env python3
"""Per-kernel sums of one rocprofv3 --pmc pass (CSV output): python scripts/pmc_kernels.py <dir> [name filter ...]"""
import collections, csv, glob, os, sys
d = sys.argv[1]; filt = sys.argv[2:]
tab = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void mqc::", "").split("(")[0]
        if filt and not any(s in k for s in filt): continue
        tab[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, v in sorted(tab.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    print(k, "launches", len(n[k]))
    for c, x in sorted(v.items()): print("   %-34s %.4e" % (c, x))
