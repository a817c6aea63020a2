import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from motifscan_amd import _lib, synth
_lib.set_device(0)
for P in (50, 579):
    vals, widths, cutoffs = synth.load_motif_set(P)
    pw = _lib.PwmSet(vals, widths, cutoffs)
    for R, L in ((1, 64), (1, 4096), (100, 500), (1000, 500), (4000, 500)):
        b, o = synth.make_regions(R, L, seed=1)
        sq = _lib.SeqSet(b, o)
        best = None
        for _ in range(8):
            r = _lib.scan(pw, sq, 3); st = r.stats(); r.close()
            t = (st["ms_prefilter"], st["ms_exact"], st["ms_sort"], st["ms_finalize"], st["ms_total"])
            best = t if best is None or t[4] < best[4] else best
        print(f"P {P} bases {R*L:8d}: prefilter {best[0]:.4f} fp64 {best[1]:.4f} sort {best[2]:.4f} finalize {best[3]:.4f} total {best[4]:.4f} ms hits {st['n_hits']}")
