#!/usr/bin/env python3
"""What is different in the occasional slow pass of the all-hits end-to-end leg?  N passes, each timed by itself; for every pass slower than
1.1 x the median: its stage clocks, the device / pinned pool deltas, and the batches whose scan device time stands out.  (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motifscan_amd import _lib, synth, dist as msdist
_lib.set_device(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
wl = synth.c4_shard(0, 1)
pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], wl["cutoffs"])
pins, batches = [], []
for k, (bases, offsets) in enumerate(wl["sets"]):
    pin = _lib.PinnedBuffer(bases.size); pin.array[:] = bases; pins.append(pin)
    for r0, r1 in msdist.batch_bounds(len(offsets) - 1, 125_000, ramp=True, max_batch=250_000, ramp_up=k == 0, ramp_down=k == 1):
        lo, hi = int(offsets[r0]), int(offsets[r1])
        batches.append((pin.array[lo:hi], np.ascontiguousarray(offsets[r0:r1 + 1] - lo)))
rows = []
for p in range(N):
    st = {}
    h0, d0 = _lib.host_pool_stats(), _lib.pool_stats()
    t0 = time.perf_counter()
    per_batch = []
    tb = t0
    for res in _lib.scan_stream(pw, iter(batches), 3, 0, depth=2, packed=12, stage_stats=st):
        s = res.stats()
        now = time.perf_counter()
        per_batch.append((round(s["ms_total"], 2), s["n_passes"], round((now - tb) * 1e3, 1)))
        tb = now
        res.close()
    ms = (time.perf_counter() - t0) * 1e3
    h1, d1 = _lib.host_pool_stats(), _lib.pool_stats()
    rows.append((ms, st, per_batch, {k: h1[k] - h0[k] for k in h0}, {k: d1[k] - d0[k] for k in ("hits", "misses", "driver_frees", "driver_ms")}))
med = sorted(r[0] for r in rows[5:])[len(rows[5:]) // 2]
print("passes:", " ".join("%.1f" % r[0] for r in rows), "| median of the last", len(rows) - 5, ":", round(med, 1))
ref = rows[-1]
print("a normal pass: stages", {k: (round(v["ms_work"], 1), round(v["ms_wait_in"], 1), round(v["ms_wait_out"], 1)) for k, v in ref[1].items()}, "batches (device ms, scan passes, ms until handed out)", ref[2])
for i, r in enumerate(rows[5:], 5):
    if r[0] > 1.1 * med:
        print(f"SLOW pass {i}: {r[0]:.1f} ms  stages", {k: (round(v["ms_work"], 1), round(v["ms_wait_in"], 1), round(v["ms_wait_out"], 1)) for k, v in r[1].items()}, "pinned", r[3], "device pool", r[4])
        print("     batches", r[2])
