#!/usr/bin/env python3
"""Is the pre-filter power-limited?  The same instruction stream on (a) the benchmark tables and (b) tables that are all
zero except one -1 per row (cutoffs > 1: no strand can hit, nothing is emitted), with candidate emission switched off in
both.  Equal cycles, different clock => the wall-time difference is the chip's DVFS answer to operand activity."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motifscan_amd import _lib, synth
_lib.set_device(0)
os.environ["MS_MEASURE"] = "1"          # opt in to the library's measurement switches
os.environ["MS_PF_CLOCK"] = "1"
os.environ["MS_PF_NOEMIT"] = os.environ.get("NOEMIT", "1")
wl = synth.workload("c4shard")
sq = _lib.SeqSet(*wl["sets"][0])
cases = [("benchmark tables", wl["cutoffs"], None), ("all-dead tables (zeros)", np.full(len(wl["cutoffs"]), 2.0), None)]
for tag, cut, bq in cases + cases:              # every case twice: order effects show
    pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], cut)
    best = None
    for _ in range(6):
        r = _lib.scan(pw, sq, 3); st = r.stats(); r.close()
        if best is None or st["ms_prefilter"] < best["ms_prefilter"]:
            best = st
    cyc = best["ms_prefilter"] * 1e-3 * best["pf_clock_mhz"] * 1e6
    print(f"{tag:26s}: pre-filter {best['ms_prefilter']:.3f} ms at {best['pf_clock_mhz']:.0f} MHz = {cyc / 1e6:.2f} M cycles, "
          f"{best['n_pwms'] - best['n_pwms_exact']} motifs on the matrix cores, {best['n_hits']} hits", flush=True)
