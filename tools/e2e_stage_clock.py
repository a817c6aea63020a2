#!/usr/bin/env python3
"""Twelve all-hits end-to-end passes, each with the stream's stage clocks (work / waits of the upload, scan and copy-out stages) and the
pool deltas: what bounds a pass.  python3 tools/e2e_stage_clock.py  (GPU box)"""
import os, sys, time
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
        batches.append((pin.array[lo:hi], np.ascontiguousarray(offsets[r0:r1 + 1] - lo)))
for p in range(12):
    st = {}
    h0, d0 = _lib.host_pool_stats(), _lib.pool_stats()
    t0 = time.perf_counter()
    for res in _lib.scan_stream(pw, iter(batches), 3, 0, depth=2, packed=int(os.environ.get("E2E_PACKED", "12")), stage_stats=st):
        res.close()
    ms = (time.perf_counter() - t0) * 1e3
    h1, d1 = _lib.host_pool_stats(), _lib.pool_stats()
    fmt = lambda s: " ".join("%s %.1f" % (k.replace("ms_", ""), v) for k, v in st[s].items() if k.startswith("ms_"))
    print("pass %2d %6.1f ms | upload: %s | scan: %s | copy_out: %s | pinned +%d malloc %.1f ms, device +%d miss %.1f ms " % (
        p, ms, fmt("upload"), fmt("scan"), fmt("copy_out"), h1["misses"] - h0["misses"], h1["ms_in_driver"] - h0["ms_in_driver"],
        d1["misses"] - d0["misses"], d1["driver_ms"] - d0["driver_ms"]), flush=True)
