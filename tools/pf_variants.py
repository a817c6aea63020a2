#!/usr/bin/env python3
"""A/B the pre-filter kernel variants in ONE process (same data, interleaved), print kernel ms.
Usage (GPU box): python tools/pf_variants.py workload v:b [v:b ...]   (variant : blocks per CU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motifscan_amd import _lib, synth

wl_name = sys.argv[1] if len(sys.argv) > 1 else "c4shard"
combos = [tuple(int(x) for x in a.split(":")) for a in sys.argv[2:]] or [(1, 1), (0, 1), (5, 2), (6, 2)]
_lib.set_device(0)
wl = synth.workload(wl_name)
n_pw = int(os.environ.get("N_PWMS", "0")) or len(wl["widths"])       # optional: first N motifs only
max_w = int(os.environ.get("MAX_W", "0"))                            # optional: only motifs of width <= MAX_W
if max_w:
    import numpy as np
    mats = synth.matrices_of(wl["pwm_values"], wl["widths"])
    keep = [i for i in range(len(mats)) if wl["widths"][i] <= max_w]
    pw = _lib.PwmSet(np.concatenate([mats[i].ravel() for i in keep]), wl["widths"][keep], wl["cutoffs"][keep])
    print(f"{len(keep)} motifs of width <= {max_w}")
else:
    pw = _lib.PwmSet(wl["pwm_values"][:4 * int(wl["widths"][:n_pw].sum())], wl["widths"][:n_pw], wl["cutoffs"][:n_pw])
sq = _lib.SeqSet(*wl["sets"][0])
os.environ["MS_MEASURE"] = "1"          # opt in to the library's measurement switches
os.environ["MS_PF_CLOCK"] = "1"
for rep in range(2):
    for noemit in (0, 1):
        for v, b in combos:
            os.environ["MS_PF_VARIANT"] = str(v)
            os.environ["MS_PF_ENGINE"] = "3" if v >= 28 else "2" if v in (24, 25) else "1" if v >= 16 else "0"       # variants >= 16: int8 matrix-core engines
            os.environ["MS_PF_BLOCKS_PER_CU"] = str(b)
            os.environ["MS_PF_NOEMIT"] = str(noemit)
            ms = []
            for _ in range(4):
                r = _lib.scan(pw, sq, 3)
                st = r.stats()
                ms.append(st["ms_prefilter"])
                r.close()
            print(f"rep {rep} variant {v} blocks/CU {b} noemit {noemit}: prefilter {min(ms):.3f} ms "
                  f"(all {['%.2f' % m for m in ms]}) tiles {st['n_tiles']} clock {st['pf_clock_mhz']:.0f} MHz "
                  f"cand {st['n_candidates']} hits {st['n_hits']}", flush=True)
