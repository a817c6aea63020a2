#!/usr/bin/env python3
"""Where the wall time of the configs[4] shard goes: extract (host loop + H2D + gather kernel), scan, frees."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motifscan_amd import _lib, synth
_lib.set_device(0)
wl = synth.workload("c5shard")
pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], wl["cutoffs"])
g = _lib.ResidentGenome({"chr": wl["genome"]})
for it in range(3):
    t0 = time.perf_counter(); sq = g.extract(*wl["windows"])
    t1 = time.perf_counter(); r = _lib.scan(pw, sq, 3)
    t2 = time.perf_counter(); st = r.stats(); rc = r.region_counts()
    t3 = time.perf_counter(); r.close(); sq.close()
    t4 = time.perf_counter()
    print(f"iter {it}: extract {1e3*(t1-t0):.1f} ms, scan wall {1e3*(t2-t1):.1f} ms (device {st['ms_total']:.1f}), counts {1e3*(t3-t2):.1f} ms, free {1e3*(t4-t3):.1f} ms", flush=True)
