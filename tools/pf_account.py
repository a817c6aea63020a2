#!/usr/bin/env python3
"""Where the pre-filter's time goes, by subtraction (measurement instantiation of the kernel, MS_MEASURE=1):
    MS_PF_NOEMIT=0  the product's work                                   MS_PF_NOEMIT=1  no candidate hand-off (events skipped)
    MS_PF_NOEMIT=3  operand reads + matrix instructions, no inspection   MS_PF_NOEMIT=2  per-pass / per-class set-up only (no row tiles)
    MS_PF_NOEMIT=4  events run and park, the entries are dropped (no decode)   MS_PF_NOEMIT=5  ... and not stored either (the events' control code alone)
each with the kernel's own clock stamps (cycles per wave and 64-window pass).  Usage (GPU box): python tools/pf_account.py [p-value] [full]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motifscan_amd import _lib, synth

pkey = sys.argv[1] if len(sys.argv) > 1 else "1e-4"
_lib.set_device(0)
wl = synth.c4_shard(0, 1) if len(sys.argv) > 2 and sys.argv[2] == "full" else synth.workload("c4shard")
vals, widths, cutoffs = synth.load_motif_set(len(wl["widths"]), pkey)
pw = _lib.PwmSet(vals, widths, cutoffs)
sq = _lib.SeqSet(*wl["sets"][0])
os.environ["MS_MEASURE"] = "1"
os.environ["MS_PF_CLOCK"] = "1"
passes_per_wave = sq.n_bases / 64 / 4096
for mode, what in ((0, "product work"), (4, "events park, nothing decoded"), (5, "events without their stores"), (1, "no hand-off"), (3, "reads + matrix instructions only"), (2, "set-up only"), (0, "product work (again)")):
    os.environ["MS_PF_NOEMIT"] = str(mode)
    best = None
    for _ in range(6):
        r = _lib.scan(pw, sq, 3)
        st = r.stats()
        r.close()
        if best is None or st["ms_prefilter"] < best[0]:
            best = (st["ms_prefilter"], st["pf_clock_mhz"])
    cyc = best[0] * 1e-3 * best[1] * 1e6 / passes_per_wave
    print(f"{sq.n_bases / 1e6:.1f} Mbase p {pkey} mode {mode} ({what}): prefilter {best[0]:.3f} ms at {best[1]:.0f} MHz = {cyc:.0f} cycles per wave and pass", flush=True)
