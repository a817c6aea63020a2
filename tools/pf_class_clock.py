#!/usr/bin/env python3
"""Where a pre-filter wave's cycles go (measurement only): MS_PF_CLOCK=2 makes the library print, per class of row tiles, the share
of wave 0's cycles spent inside it.  Usage (GPU box): python tools/pf_class_clock.py [strand] [p-value key] [full]
("full": one 500-Mbase set of configs[3], the launch bench.py times; default: the 62.5-Mbase shard)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motifscan_amd import _lib, synth

strand = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pkey = sys.argv[2] if len(sys.argv) > 2 else "1e-4"
_lib.set_device(0)
wl = synth.c4_shard(0, 1) if len(sys.argv) > 3 and sys.argv[3] == "full" else synth.workload("c4shard")
vals, widths, cutoffs = synth.load_motif_set(len(wl["widths"]), pkey)
pw = _lib.PwmSet(vals, widths, cutoffs)
sq = _lib.SeqSet(*wl["sets"][0])
os.environ["MS_MEASURE"] = "1"
for noemit in (1, 0):
    os.environ["MS_PF_NOEMIT"] = str(noemit)
    os.environ["MS_PF_CLOCK"] = "1"
    for _ in range(3):
        _lib.scan(pw, sq, strand).close()
    os.environ["MS_PF_CLOCK"] = "2"
    r = _lib.scan(pw, sq, strand)
    st = r.stats()
    print(f"{sq.n_bases / 1e6:.1f} Mbase, strand mask {strand} p {pkey} noemit {noemit}: prefilter {st['ms_prefilter']:.3f} ms clock {st['pf_clock_mhz']:.0f} MHz", flush=True)
    r.close()
