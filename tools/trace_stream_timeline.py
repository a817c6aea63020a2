import csv, glob, sys, collections
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:44], r.get("Queue_Id", "?")))
cols = None
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cols = list(r.keys())
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?").replace("MEMORY_COPY_", ""), r.get("Stream_Id", "?")))
print("copy columns:", cols)
ev.sort()
t0 = ev[0][0]
# the pipelined pass: locate the last 40 pack_kernel launches (the e2e passes run after the resident steps)
packs = [i for i, e in enumerate(ev) if "pack_kernel" in e[2]]
print("pack_kernel launches", len(packs))
agg = collections.defaultdict(list)
for s, e, n, q in ev:
    agg[n].append((e - s) / 1e6)
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:16]:
    big = [x for x in v if x > 0.05]
    print("%-52s n=%5d total %9.2f ms   >50us: n=%4d avg %7.3f max %7.3f" % (n, len(v), sum(v), len(big), (sum(big) / len(big)) if big else 0, max(v)))
# timeline excerpt: 3 batches in the middle of the first pipelined pass (packs 20..23 of the e2e region)
ph = [i for i, e in enumerate(ev) if "pack_hits" in e[2]]
i0 = ph[len(ph) * 3 // 4] if ph else packs[len(packs) // 2]
tb = ev[i0][0]
for s, e, n, q in ev[i0:]:
    if (s - tb) / 1e6 > 10: break
    if (e - s) > 30000:
        print("%9.3f  +%7.3f  %-50s q/s %s" % ((s - tb) / 1e6, (e - s) / 1e6, n, q))
