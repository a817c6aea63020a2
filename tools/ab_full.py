#!/usr/bin/env python3
"""Stage times of ONE full-size configs[3] set (1M regions x 500 bp = 500 Mbase, 579 PWMs, both strands) through the product path
(no measurement switches), for same-box A/B runs of two builds: python tools/ab_full.py [strand] [p-value key]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motifscan_amd import _lib, synth

strand = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pkey = sys.argv[2] if len(sys.argv) > 2 else "1e-4"
_lib.set_device(0)
wl = synth.c4_shard(0, 1)
vals, widths, cutoffs = synth.load_motif_set(len(wl["widths"]), pkey)
pw = _lib.PwmSet(vals, widths, cutoffs)
sq = _lib.SeqSet(*wl["sets"][0])
rows = []
for i in range(8):
    r = _lib.scan(pw, sq, strand)
    st = r.stats()
    rows.append((st["ms_prefilter"], st["ms_exact"], st["ms_sort"], st["ms_finalize"], st["ms_total"]))
    r.close()
best = min(rows[2:])
print(f"{sq.n_bases / 1e6:.0f} Mbase strand {strand} p {pkey}: prefilter {best[0]:.2f} fp64 {best[1]:.2f} sort {best[2]:.2f} finalize {best[3]:.2f} total {best[4]:.2f} ms "
      f"(prefilter of the last 6: {['%.2f' % x[0] for x in rows[2:]]})", flush=True)
