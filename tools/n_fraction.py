#!/usr/bin/env python3
"""What non-ACGT bases cost (they are all-zero one-hot columns of the same pre-filter: no separate path): the same 250k x 500 bp x 579
scan with 0 %, 1 % (the benchmark's) and 5 % of the regions holding a run of 1-50 N; then one 125-Mbase chromosome with assembly gaps
(runs of 1e4 ... 5e5 N) covering 0 %, 10 % and 30 % of it.  Prints the stage times of ms_scan."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motifscan_amd import _lib, synth
_lib.set_device(0)
vals, widths, cutoffs = synth.load_motif_set(579)
pw = _lib.PwmSet(vals, widths, cutoffs)
for frac in (0.0, 0.01, 0.05):
    bases, offsets = synth.make_regions(250_000, 500, seed=3, frac_n=frac)
    sq = _lib.SeqSet(bases, offsets)
    best = None
    for _ in range(8):
        r = _lib.scan(pw, sq, 3); st = r.stats(); r.close()
        if best is None or st["ms_total"] < best["ms_total"]:
            best = st
    print(f"frac_n {frac:.2f}: pre-filter {best['ms_prefilter']:.3f}  fp64 stage {best['ms_exact']:.3f}  sort {best['ms_sort']:.3f}  finalize {best['ms_finalize']:.3f}  "
          f"total {best['ms_total']:.3f} ms; {best['n_candidates']} candidates, {best['n_hits']} hits", flush=True)
    sq.close()
for gap_share in (0.0, 0.1, 0.3):
    L = 125_000_000
    g, _ = synth.make_regions(1, L, seed=5, frac_n=0.0)
    rng = np.random.default_rng(6)
    covered = 0
    while covered < gap_share * L:
        n = int(rng.integers(10_000, 500_001))
        st = int(rng.integers(0, L - n))
        g[st:st + n] = ord("N")
        covered += n
    sq = _lib.SeqSet(g, np.array([0, L], dtype=np.int64))
    best = None
    for _ in range(6):
        r = _lib.scan(pw, sq, 3); st_ = r.stats(); r.close()
        if best is None or st_["ms_total"] < best["ms_total"]:
            best = st_
    print(f"one chromosome, {100 * float((g == ord('N')).mean()):.1f} % of it in gaps: pre-filter {best['ms_prefilter']:.3f}  fp64 stage {best['ms_exact']:.3f}  sort {best['ms_sort']:.3f}  "
          f"total {best['ms_total']:.3f} ms; {best['n_candidates']} candidate slots, {best['n_hits']} hits", flush=True)
    sq.close()
