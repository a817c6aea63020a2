#!/usr/bin/env python3
"""PCIe-inclusive wall time of one scan as a Python user sees it: host ASCII in -> numpy hit arrays out."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motifscan_amd import _lib, synth
_lib.set_device(0)
wl = synth.workload(sys.argv[1] if len(sys.argv) > 1 else "c4shard")
bases, offsets = wl["sets"][0]
pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], wl["cutoffs"])
units = int(offsets[-1]) * wl["n_pwms"]
for it in range(4):
    t0 = time.perf_counter(); sq = _lib.SeqSet(bases, offsets)
    t1 = time.perf_counter(); r = _lib.scan(pw, sq, 3)
    t2 = time.perf_counter(); h = r.hits(copy=False)
    t3 = time.perf_counter(); h2 = {k: v.copy() for k, v in h.items()}
    t4 = time.perf_counter(); r.close(); sq.close()
    t5 = time.perf_counter()
    print(f"iter {it}: H2D+pack {1e3*(t1-t0):.1f} ms, scan {1e3*(t2-t1):.1f} ms, hits->pinned views {1e3*(t3-t2):.1f} ms "
          f"({len(h['pos'])} hits), numpy copies {1e3*(t4-t3):.1f} ms, free {1e3*(t5-t4):.1f} ms | "
          f"end-to-end (views) {units/(t3-t0):.3e} U/s", flush=True)
