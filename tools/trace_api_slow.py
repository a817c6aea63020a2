#!/usr/bin/env python3
"""From a rocprofv3 --hip-runtime-trace (+ --kernel-trace --memory-copy-trace) CSV directory of tools/trace_e2e_pass.py: the passes (from the program's
own log), and for every pass the HIP API calls that took more than 1.5 ms, per host thread.  Usage: trace_api_slow.py DIR"""
import csv, glob, sys, collections
d = sys.argv[1]
api = []
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r["Thread_Id"]))
api.sort()
pf = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "prefilter_f6" in r["Kernel_Name"]: pf.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
pf.sort()
per = 14
print("api calls:", len(api), "prefilter launches:", len(pf))
for p in range(len(pf) // per):
    t0, t1 = pf[p * per][0], pf[p * per + per - 1][1]
    span = (t1 - t0) / 1e6
    slow = [(e - s, fn, th, s) for s, e, fn, th in api if s >= t0 - 3e6 and s <= t1 and e - s > 1.5e6]
    tot = collections.Counter()
    for dur, fn, th, s in slow: tot[(th, fn)] += dur / 1e6
    print("pass %d: first to last pre-filter %.1f ms; calls > 1.5 ms: %s" % (p, span, ", ".join("%s/%s %.1f" % (k[0][-4:], k[1], v) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:8])))
