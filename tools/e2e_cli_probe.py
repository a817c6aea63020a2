#!/usr/bin/env python3
"""Why is the reference CLI's job (input sites out, control set counted only) erratic end to end?  Twelve passes of the cli leg and of the all-hits leg,
each pass timed by itself, with the stream's stage clocks of every pass.  python tools/e2e_cli_probe.py [variant]  (GPU box)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motifscan_amd import _lib, synth, dist as msdist
_lib.set_device(0)
wl = synth.c4_shard(0, 1)
pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], wl["cutoffs"])
pins, batches = [], []
for k, (bases, offsets) in enumerate(wl["sets"]):
    pin = _lib.PinnedBuffer(bases.size); pin.array[:] = bases; pins.append(pin)
    for r0, r1 in msdist.batch_bounds(len(offsets) - 1, 125_000, ramp=True, max_batch=250_000, ramp_up=k == 0, ramp_down=k == 1):
        lo, hi = int(offsets[r0]), int(offsets[r1])
        batches.append((pin.array[lo:hi], np.ascontiguousarray(offsets[r0:r1 + 1] - lo), k > 0))
def run(cli, read_counts=True):
    st = {}
    t0 = time.perf_counter()
    n = 0
    bl = batches if cli else [(b, o) for b, o, _ in batches]
    for res in _lib.scan_stream(pw, iter(bl), 3, 0, depth=2, packed=12, stage_stats=st):
        n += res.n_hits
        if cli and read_counts:
            res.region_counts()
        res.close()
    return (time.perf_counter() - t0) * 1e3, st
for name, cli, rc in (("all hits", False, False), ("cli, counts read", True, True), ("cli, counts not read", True, False), ("all hits", False, False), ("cli, counts read", True, True), ("all hits", False, False), ("all hits", False, False)):
    h0, d0 = _lib.host_pool_stats(), _lib.pool_stats()
    rows = [run(cli, rc) for _ in range(10)]
    h1, d1 = _lib.host_pool_stats(), _lib.pool_stats()
    print("   pinned pool: +%d served, +%d hipHostMalloc, +%d hipHostFree, +%.1f ms in the driver | device pool: %s -> %s" % (h1["hits"] - h0["hits"], h1["misses"] - h0["misses"], h1["driver_frees"] - h0["driver_frees"], h1["ms_in_driver"] - h0["ms_in_driver"], d0, d1), flush=True)
    ms = [r[0] for r in rows]
    print(f"{name:24s} passes {' '.join('%.1f' % x for x in ms)}  | median {sorted(ms)[5]:.1f}  | last pass scan work {rows[-1][1]['scan']['ms_work']:.1f} wait_in {rows[-1][1]['scan']['ms_wait_in']:.1f} upload {rows[-1][1]['upload']['ms_work']:.1f} copy_out {rows[-1][1]['copy_out']['ms_work']:.1f}", flush=True)
