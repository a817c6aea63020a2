#!/usr/bin/env python3
"""What the N path costs: the pre-filter stage (matrix-core kernel + nwin_mark/nwin_scan) of one 62.5-Mbase set whose regions
hold runs of N in a growing fraction of the regions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motifscan_amd import _lib, synth
_lib.set_device(0)
vals, widths, cutoffs = synth.load_motif_set(579)
pw = _lib.PwmSet(vals, widths, cutoffs)
for frac in (0.0, 0.001, 0.01, 0.05, 0.2):
    bases, offsets = synth.make_regions(125_000, 500, seed=5, frac_n=frac)
    sq = _lib.SeqSet(bases, offsets)
    best = None
    for _ in range(5):
        r = _lib.scan(pw, sq, 3); st = r.stats(); r.close()
        if best is None or st["ms_prefilter"] < best["ms_prefilter"]:
            best = st
    print(f"regions with a run of N: {100 * frac:5.1f} %   pre-filter stage {best['ms_prefilter']:.3f} ms   fp64 {best['ms_exact']:.3f} ms   records {best['n_candidates']}  hits {best['n_hits']}", flush=True)
    sq.close()
