#!/usr/bin/env python3
"""Device time of a counts-only scan against the ordered scan, one 250 000-region batch and one full 1M-region set (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motifscan_amd import _lib, synth
_lib.set_device(0)
wl = synth.c4_shard(0, 1)
pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], wl["cutoffs"])
bases, offsets = wl["sets"][0]
for n in (250_000, 1_000_000):
    sq = _lib.SeqSet(bases[:int(offsets[n])], offsets[:n + 1])
    for flags, name in ((0, "ordered"), (_lib.MS_SCAN_COUNTS_ONLY, "counts only")):
        rows = []
        for _ in range(6):
            r = _lib.scan(pw, sq, 3, flags); st = r.stats(); rows.append((st["ms_total"], st["ms_prefilter"], st["ms_exact"], st["ms_sort"], st["ms_finalize"])); c = r.region_counts(); r.close()
        b = min(rows[2:])
        print(f"{n:8d} regions {name:12s} total {b[0]:.2f} ms  prefilter {b[1]:.2f} fp64 {b[2]:.2f} order/count {b[3]:.2f} finalize {b[4]:.2f}   counts sum {int(c.sum())}", flush=True)
    sq.close()
