#!/usr/bin/env python3
"""How much of the pre-filter's LDS idle time is the per-class structure?  Scan with the benchmark
motif set, and with synthetic sets whose motifs all have ONE width (one class)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motifscan_amd import _lib, synth
_lib.set_device(0)
os.environ["MS_MEASURE"] = "1"          # opt in to the library's measurement switches
os.environ["MS_PF_CLOCK"] = "1"
vals, widths, cutoffs = synth.load_motif_set(579)
mats = synth.matrices_of(vals, widths)
bases, offsets = synth.make_regions(125_000, 500, seed=1)
sq = _lib.SeqSet(bases, offsets)

def run(tag, pw):
    for noemit in (0, 1):
        os.environ["MS_PF_NOEMIT"] = str(noemit)
        best = None
        for _ in range(4):
            r = _lib.scan(pw, sq, 3); st = r.stats(); r.close()
            if best is None or st["ms_prefilter"] < best["ms_prefilter"]:
                best = st
        lds = best["lds_bytes_read"] / (best["ms_prefilter"] * 1e-3)
        line = (f"{tag:28s} noemit {noemit}: {best['ms_prefilter']:.3f} ms, tiles {best['n_tiles']}, cand {best['n_candidates']}, "
                f"LDS {lds/1e12:.1f} TB/s = {100*lds/(256*256*best['pf_clock_mhz']*1e6):.1f}% at {best['pf_clock_mhz']:.0f} MHz")
        if best["pf_engine"] == 3:                                   # cycles per matrix instruction (32x32x64: 131072 ops) per SIMD
            n_mfma = best["mfma_ops"] / 131072 / 1024
            line += f", {best['ms_prefilter'] * 1e-3 * best['pf_clock_mhz'] * 1e6 / n_mfma:.1f} cycles per MFMA per SIMD"
        print(line, flush=True)

only = [int(x) for x in os.environ.get("PF_WIDTHS", "8,12,15,16,21").split(",")]
run("benchmark set", _lib.PwmSet(vals, widths, cutoffs))
for W in only:
    sel = [i for i in range(579) if widths[i] == W]
    reps = (579 + len(sel) - 1) // len(sel)
    idx = (sel * reps)[:579]
    pw = _lib.PwmSet.from_matrices([mats[i] for i in idx], cutoffs[idx])
    run(f"all W={W} (one class)", pw)
