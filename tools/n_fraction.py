#!/usr/bin/env python3
"""What the non-ACGT path costs: the same 250k x 500 bp x 579 scan with 0 %, 1 % (the benchmark's) and 5 % of the regions
holding a run of 1-50 N.  Prints the stage times of ms_scan."""
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
