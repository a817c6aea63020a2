#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace --memory-copy-trace CSV directory: the busy time of the scan's kernels in the LAST batch-stream
pass, the idle gaps between them, and what ran in the gaps.  Usage: trace_gaps.py DIR"""
import csv, glob, sys, collections
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], "K"))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?").replace("MEMORY_COPY_", ""), "C"))
ev.sort()
pf = [e for e in ev if "prefilter_f6" in e[2]]
print("prefilter launches:", len(pf))
# tools/e2e_stages.py: 4 passes of 20 batches (ramp, packed), 4 x 20 (counts only), 4 x 16 (equal batches): take one pass by launch index
first = int(sys.argv[2]) if len(sys.argv) > 2 else 60
count = int(sys.argv[3]) if len(sys.argv) > 3 else 20
run = pf[first:first + count]
t0, t1 = run[0][0] - 3e6, run[-1][1] + 8e6
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
scan_names = ("prefilter_f6", "rescore", "fill_tail", "onesweep", "radix", "finalize", "pack_hits", "exact_all", "histogram", "sort")
busy = sorted([(s, e) for s, e, n, k in win if k == "K"])
tot = collections.defaultdict(float)
for s, e, n, k in win:
    tot[n] += (e - s) / 1e6
print("window %.2f ms" % ((t1 - t0) / 1e6))
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]:
    print("   %-60s %8.2f ms" % (n, v))
# union of kernel-busy intervals
merged = []
for s, e in busy:
    if merged and s <= merged[-1][1]: merged[-1][1] = max(merged[-1][1], e)
    else: merged.append([s, e])
busy_ms = sum(e - s for s, e in merged) / 1e6
print("kernels busy (union) %.2f ms of %.2f ms between first and last kernel" % (busy_ms, (merged[-1][1] - merged[0][0]) / 1e6))
gaps = [(merged[i + 1][0] - merged[i][1], merged[i][1]) for i in range(len(merged) - 1)]
big = sorted([g for g in gaps if g[0] > 30e3], reverse=True)
print("idle gaps > 30 us: n=%d total %.2f ms; largest:" % (len(big), sum(g[0] for g in big) / 1e6))
for g, at in big[:12]:
    before = [n for s, e, n, k in win if k == "K" and abs(e - at) < 2e3]
    after = [n for s, e, n, k in win if k == "K" and abs(s - (at + g)) < 2e3]
    cp = [n for s, e, n, k in win if k == "C" and s < at + g and e > at]
    print("   %.3f ms at +%.2f ms after %s before %s copies %s" % (g / 1e6, (at - t0) / 1e6, before[:1], after[:1], cp[:2]))
# timeline of the first 6 ms
print("timeline (events > 40 us) of 8 ms from the middle:")
mid = t0 + (t1 - t0) / 2
for s, e, n, k in win:
    if mid <= s <= mid + 8e6 and e - s > 40e3:
        print("   %8.3f +%7.3f %s" % ((s - mid) / 1e6, (e - s) / 1e6, n))
