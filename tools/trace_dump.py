#!/usr/bin/env python3
"""Dump kernels of a rocprofv3 kernel trace in a time window around the k-th prefilter launch: name, queue, grid, duration."""
import csv, glob, sys
d = sys.argv[1]; k = int(sys.argv[2]) if len(sys.argv) > 2 else 70
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        r["Kernel_Name"] = "COPY " + r.get("Direction", "?") + " " + str(r.get("Bytes", ""))
        rows.append(r)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pf = [r for r in rows if "prefilter_f6" in r["Kernel_Name"]]
t0 = int(pf[k]["Start_Timestamp"]) - 1_500_000
print("columns:", [c for c in pf[0].keys()][:20])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 <= s <= t0 + 7_000_000:
        print("%8.3f +%7.3f q%-3s grid %-10s wg %-5s %s" % ((s - t0) / 1e6, (e - s) / 1e6, r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")), r["Kernel_Name"].split("(")[0][:70]))
