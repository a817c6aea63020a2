#!/usr/bin/env python3
"""What would the pre-filter take if the benchmark set's row tiles had the shape a 3-slot ("reference base") operand encoding gives
them?  With 3 k-slots per column a half-block holds 10 columns instead of 7 + bias: motifs of <= 10 columns cost one half-block,
11 ... 20 two.  This probe scans full-size configs[3] sets (500 Mbase) with
    (a) the benchmark set as it is,
    (b) a SHAPE-EQUIVALENT set for the CURRENT planner: every motif of <= 10 columns replaced by one of the set's own 7-column
        motifs, every motif of 11 ... 20 columns by one of its 15-column motifs (their own p-value cutoffs: the same hit density per
        motif), wider motifs as they are -- the plan then has the class structure (8 + 10 paired row tiles, 2 plain) and the
        matrix-instruction count (64 per pass instead of 82) the 3-slot plan would have for the real set.
It is a timing probe (kill criterion for building the encoding), not a parity vector.  python tools/pf_shape_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motifscan_amd import _lib, synth

_lib.set_device(0)
wl = synth.c4_shard(0, 1)
vals, widths, cutoffs = synth.load_motif_set(579)
mats = synth.matrices_of(vals, widths)
sq = _lib.SeqSet(*wl["sets"][0])


def run(tag, pw):
    rows = []
    for i in range(8):
        r = _lib.scan(pw, sq, 3)
        st = r.stats()
        rows.append((st["ms_prefilter"], st["ms_exact"], st["ms_sort"], st["ms_finalize"], st["ms_total"], st["n_candidates"], st["n_hits"], st["mfma_ops"]))
        r.close()
    b = min(rows[2:])
    print(f"{tag:34s} prefilter {b[0]:6.2f} fp64 {b[1]:5.2f} sort {b[2]:5.2f} finalize {b[3]:5.2f} total {b[4]:6.2f} ms  candidates {b[5]} hits {b[6]} "
          f"matrix instructions per 64 windows {b[7] / 131072 / (sq.n_bases / 64):.1f}", flush=True)


def pick(width, k):
    sel = [i for i in range(579) if widths[i] == width]
    return sel[k % len(sel)]


run("benchmark set", _lib.PwmSet(vals, widths, cutoffs))
idx = []
for i, w in enumerate(widths):
    idx.append(pick(7, i) if w <= 10 else (pick(15, i) if w <= 20 else i))
run("3-slot shape (7 / 15 / as is)", _lib.PwmSet.from_matrices([mats[i] for i in idx], cutoffs[idx]))
idx9 = []
for i, w in enumerate(widths):
    idx9.append(pick(7, i) if w <= 9 else (pick(15, i) if w <= 20 else i))
run("3-slot shape, 9 columns per half", _lib.PwmSet.from_matrices([mats[i] for i in idx9], cutoffs[idx9]))
run("benchmark set (again)", _lib.PwmSet(vals, widths, cutoffs))
