#!/usr/bin/env python3
"""A/B the pre-filter kernel variants in ONE process (same data, interleaved), print kernel ms.
Usage (GPU box): python tools/pf_variants.py [workload] [variants...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motifscan_amd import _lib, synth

wl_name = sys.argv[1] if len(sys.argv) > 1 else "c4shard"
variants = [int(v) for v in sys.argv[2:]] or [0, 1, 2, 3, 4]
_lib.set_device(0)
wl = synth.workload(wl_name)
pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], wl["cutoffs"])
sq = _lib.SeqSet(*wl["sets"][0])
for rep in range(2):
    for noemit in (0, 1):
        for v in variants:
            os.environ["MS_PF_VARIANT"] = str(v)
            os.environ["MS_PF_NOEMIT"] = str(noemit)
            ms = []
            for _ in range(4):
                r = _lib.scan(pw, sq, 3)
                st = r.stats()
                ms.append(st["ms_prefilter"])
                r.close()
            print(f"rep {rep} variant {v} noemit {noemit}: prefilter {min(ms):.3f} ms (min of 4, all {['%.2f' % m for m in ms]}) "
                  f"cand {st['n_candidates']} hits {st['n_hits']} exact {st['ms_exact']:.2f} ms", flush=True)
