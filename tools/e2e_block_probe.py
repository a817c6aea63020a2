#!/usr/bin/env python3
"""Do the slow end-to-end passes go with particular device blocks?  Ten all-hits passes with MS_MEASURE=1 MS_TRACE_BLOCKS=1 (the library prints, per scan,
the blocks it worked on and its stage times); then per pass: its time, and per batch the sequence block / result block and the fp64 stage's time.
python3 tools/e2e_block_probe.py  (GPU box; runs itself as a child with the switches set)"""
import os, sys, subprocess, re, collections
if os.environ.get("MS_TRACE_BLOCKS") != "1":
    env = dict(os.environ, MS_MEASURE="1", MS_TRACE_BLOCKS="1")
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "trace_e2e_pass.py")], env=env, capture_output=True, text=True)
    launches, dones, passes = [], [], []
    for line in (out.stderr + out.stdout).splitlines():
        if line.startswith("MSBLK launch"):
            launches.append(dict(re.findall(r"(\w+)=(0x[0-9a-f]+)", line)))
        elif line.startswith("MSBLK done"):
            d = dict(re.findall(r"(\w+)=([0-9a-fx.]+)", line)); dones.append(d)
        elif line.startswith("pass"):
            passes.append(line)
    print("\n".join(passes))
    print(len(launches), "launches,", len(dones), "completions")
    by_res = collections.defaultdict(list)
    for l in launches: by_res[l["res"]].append(l)
    # completions come in launch order per result block; join in order
    idx = collections.Counter()
    rows = []
    for d in dones:
        r = d["res"]; l = by_res[r][idx[r]] if idx[r] < len(by_res[r]) else {}; idx[r] += 1
        rows.append((l.get("codes"), r, int(d["bases"]), float(d["pf"]), float(d["fp64"]), float(d["sort"]), float(d["fin"])))
    per_pass = 14
    for p in range(len(rows) // per_pass):
        rr = rows[p * per_pass:(p + 1) * per_pass]
        print("pass %d: pf %.1f fp64 %.1f sort %.1f fin %.1f" % (p, sum(r[3] for r in rr), sum(r[4] for r in rr), sum(r[5] for r in rr), sum(r[6] for r in rr)))
    print("the fp64 stage per (sequence block, Mbases): time in ms over all passes")
    agg = collections.defaultdict(list)
    for r in rows[per_pass * 2:]: agg[(r[0], r[2] // 1000000)].append(r[4])
    for k, v in sorted(agg.items(), key=lambda kv: (kv[0][1], str(kv[0][0]))): print("   codes %s  %4d Mbases  n %2d  fp64 ms %s" % (k[0], k[1], len(v), " ".join("%.2f" % x for x in v)))
    agg = collections.defaultdict(list)
    for r in rows[per_pass * 2:]: agg[(r[1], r[2] // 1000000)].append(r[4])
    print("the fp64 stage per (result block, Mbases)")
    for k, v in sorted(agg.items(), key=lambda kv: (kv[0][1], str(kv[0][0]))): print("   res %s  %4d Mbases  n %2d  fp64 ms %s" % (k[0], k[1], len(v), " ".join("%.2f" % x for x in v)))
