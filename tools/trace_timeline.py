#!/usr/bin/env python3
"""Print a merged kernel + memory-copy timeline from a rocprofv3 --kernel-trace --memory-copy-trace CSV directory.
Usage: trace_timeline.py DIR [t_begin_ms t_end_ms]   (times relative to the first record)"""
import csv, glob, sys

d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s" % (r.get("Queue_Id", "?"), r["Kernel_Name"][:60]), r.get("Workgroup_Size", "") + "x" + r.get("Grid_Size", "")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        b = int(r.get("Bytes", 0) or 0) if "Bytes" in r else 0
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s" % r.get("Direction", "?"), "%d B" % b if b else str({k: v for k, v in r.items() if k not in ("Start_Timestamp", "End_Timestamp")})[:120]))
ev.sort()
t0 = ev[0][0]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
for s, e, n, x in ev:
    t = (s - t0) / 1e6
    if lo <= t <= hi and (e - s) > 20000:
        print("%10.3f ms  +%8.3f ms  %-70s %s" % (t, (e - s) / 1e6, n, x))
print("total span %.1f ms, %d records" % ((ev[-1][1] - t0) / 1e6, len(ev)))
