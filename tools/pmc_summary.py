#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes: per kernel and counter, the mean over launches.
Usage: python tools/pmc_summary.py <rocprof output dir> <out.csv>     (reads every *counter_collection.csv below the dir)"""
import csv, glob, os, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    per_dispatch = defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per_dispatch[key] += float(r["Counter_Value"])          # a counter is reported per XCD / instance: sum them
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for (d, cn), v in per_dispatch.items():
        k = names[d].split("(")[0].replace("void ", "")
        acc[(k, cn)][0] += 1
        acc[(k, cn)][1] += v
with open(dst, "w") as fh:
    fh.write("kernel,counter,launches,mean_per_launch\n")
    for (k, cn), (n, tot) in sorted(acc.items()):
        fh.write(f"\"{k}\",{cn},{n},{tot / n:.1f}\n")
print("wrote", dst, len(acc), "rows")
