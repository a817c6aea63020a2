#!/usr/bin/env python3
"""
CPU experiment (round 4, VERDICT r3 item 3's probe with a kill criterion): RIGOROUS COLUMN TRUNCATION of the pre-filter rows.
Keep only the K most informative columns of a motif (7 of the motifs with 8 ... 15 columns: one half-block instead of two; 15 of
those with 16 ... 23: a paired row of two half-blocks instead of a plain row of two k-blocks) and let every dropped column
contribute its BEST case (deficit 0): still an upper bound of the score, so still no false negative -- and fewer matrix
instructions per 32 windows.  What it costs is selectivity.  This script takes the shipped operand image of the benchmark motif
set (PwmSet.plan: the rows the kernel multiplies, in levels of 1/8), computes the exact candidate probability of every
(motif, strand) row on the benchmark's background by convolving the per-column level distributions, once with all columns and once
truncated, and prints candidates per hit and the instruction count either way.

Kill criterion (set before the run): build the truncated plan only if candidates per hit stay below 2.5.

    python tools/attic/truncation_sim.py [p-value key]        (no GPU: the plan is host code)
"""
import sys
import numpy as np
sys.path.insert(0, ".")
from motifscan_amd import _lib, synth

pkey = sys.argv[1] if len(sys.argv) > 1 else "1e-4"
vals, widths, cutoffs = synth.load_motif_set(579, pkey, sys.argv[2] if len(sys.argv) > 2 else "benchmark")
pw = _lib.PwmSet(vals, widths, cutoffs)
plan = pw.plan(3)
BG = synth.BG
gf, rows, bias, cols = plan["group_fields"], plan["rows"], plan["bias"], plan["group_cols"]


def cand_prob(v, b0, keep):
    """P(b0 + sum_c v[c][base_c] >= 0) with the columns NOT in `keep` at their best case.  v: [W][4] levels."""
    lo = int(v.min(axis=1).clip(max=0).sum()) + min(b0, 0) - 8
    off = -lo
    size = off + int(v.max(axis=1).clip(min=0).sum()) + max(b0, 0) + 9
    dist = np.zeros(size)
    dist[off + b0 + int(sum(v[c].max() for c in range(len(v)) if c not in keep))] = 1.0
    for c in keep:
        new = np.zeros(size)
        for b in range(4):
            s = int(v[c][b])
            if s >= 0:
                new[s:] += BG[b] * dist[:size - s]
            else:
                new[:s] += BG[b] * dist[-s:]
        dist = new
    return float(dist[off:].sum())


tot = {"full": 0.0, "trunc": 0.0}
by_class = {}
n_rows = 0
for g in range(gf.shape[0]):
    for f in range(16):
        m = int(gf[g, f])
        if m < 0:
            continue
        W = int(widths[m])
        ncol = int(cols[g]) - 1                                   # the last column of the field is the bias column
        v = rows[g, f, :W, :].astype(np.int64)                    # [W][4]
        b0 = int(bias[g, f])
        full = cand_prob(v, b0, list(range(W)))
        info = np.array([(v[c].max() - v[c]) @ BG for c in range(W)])      # expected deficit of a background base: the column's information
        if 8 <= W <= 15:
            K, cls = 7, "8..15 -> 7 columns"
        elif 16 <= W <= 23:
            K, cls = 15, "16..23 -> 15 columns"
        else:
            K, cls = W, ("<= 7 (unchanged)" if W <= 7 else ">= 24 (unchanged)")
        keep = sorted(np.argsort(-info)[:K].tolist())
        tr = cand_prob(v, b0, keep) if K < W else full
        tot["full"] += full
        tot["trunc"] += tr
        a = by_class.setdefault(cls, [0, 0.0, 0.0])
        a[0] += 1; a[1] += full; a[2] += tr
        n_rows += 1

# hits per window and row on the benchmark (measured: 61.7 M hits per 500 Mbase scan of 579 motifs x 2 strands, profiles/archive/r04a bench line)
hit_rate = {"1e-4": 61_748_087 / 500e6, "1e-3": 3.8e8 / 500e6}.get(pkey)
print(f"p = {pkey}: {n_rows} (motif, strand) rows; candidates per window, all rows: shipped rows {tot['full']:.4f}, truncated rows {tot['trunc']:.4f}")
if hit_rate:
    print(f"hits per window (measured on the benchmark regions): {hit_rate:.4f}  ->  candidates per hit: shipped {tot['full'] / hit_rate:.2f} "
          f"(device counter: 79.1 M / 61.7 M = 1.28 at 1e-4), truncated {tot['trunc'] / hit_rate:.2f}")
for cls, (n, a, b) in sorted(by_class.items()):
    print(f"  rows of motifs with {cls:24s}: {n:4d} rows, candidates per window {a:.5f} -> {b:.5f}  (x {b / max(a, 1e-30):.1f})")
w = np.asarray(widths)
n7, n15, n23, nw = int((w <= 7).sum()), int(((w >= 8) & (w <= 15)).sum()), int(((w >= 16) & (w <= 23)).sum()), int((w >= 24).sum())
ship = -(-n7 // 32) * 1 + -(-n15 // 32) * 2 + -(-(n23 + nw) // 16) * 2
trn = -(-(n7 + n15) // 32) * 1 + -(-n23 // 32) * 2 + -(-nw // 16) * 2
print(f"matrix instructions per 32 windows (row tiles x blocks, whole tiles): shipped ~{ship}, truncated ~{trn}  ({n7} / {n15} / {n23} / {nw} motifs of <= 7 / 8..15 / 16..23 / >= 24 columns)")
