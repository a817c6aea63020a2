#!/usr/bin/env python3
"""Eight passes of the all-hits end-to-end leg (configs[3], 14 batches, 12-byte copy-out) and nothing else: the program to put under
`rocprofv3 --kernel-trace --memory-copy-trace` for tools/trace_gaps.py / trace_timeline.py.  python3 tools/trace_e2e_pass.py [cli]  (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motifscan_amd import _lib, synth, dist as msdist
cli = len(sys.argv) > 1 and sys.argv[1] == "cli"
_lib.set_device(0)
wl = synth.c4_shard(0, 1)
pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], wl["cutoffs"])
pins, batches = [], []
for k, (bases, offsets) in enumerate(wl["sets"]):
    pin = _lib.PinnedBuffer(bases.size); pin.array[:] = bases; pins.append(pin)
    for r0, r1 in msdist.batch_bounds(len(offsets) - 1, 125_000, ramp=True, max_batch=250_000, ramp_up=k == 0, ramp_down=k == 1):
        lo, hi = int(offsets[r0]), int(offsets[r1])
        batches.append((pin.array[lo:hi], np.ascontiguousarray(offsets[r0:r1 + 1] - lo), k > 0) if cli else (pin.array[lo:hi], np.ascontiguousarray(offsets[r0:r1 + 1] - lo)))
for p in range(8):
    t0 = time.perf_counter()
    n = 0
    for res in _lib.scan_stream(pw, iter(batches), 3, 0, depth=2, packed=12):
        n += res.n_hits
        res.close()
    print("pass %d: %.1f ms, %d hits" % (p, (time.perf_counter() - t0) * 1e3, n), flush=True)
