#!/usr/bin/env python3
"""Where a pass of the batch stream spends its wall time outside the scan stage: stream set-up, the first batch's upload, the last
batch's copy-out, tear-down -- time stamps around the phases of `_lib.scan_stream`'s loop on the bench's own batches (configs[3])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motifscan_amd import _lib, synth, dist as msdist
_lib.set_device(0)
wl = synth.c4_shard(0, 1)
pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], wl["cutoffs"])
pins, batches = [], []
n_sets = len(wl["sets"])
for k, (bases, offsets) in enumerate(wl["sets"]):
    pin = _lib.PinnedBuffer(max(bases.size, 1)); pin.array[:bases.size] = bases; pins.append(pin)
    for r0, r1 in msdist.batch_bounds(len(offsets) - 1, 125000, ramp=True, max_batch=250000, ramp_up=k == 0, ramp_down=k == n_sets - 1):
        lo, hi = int(offsets[r0]), int(offsets[r1])
        batches.append((pin.array[lo:hi], np.ascontiguousarray(offsets[r0:r1 + 1] - lo)))
for rep in range(5):
    t0 = time.perf_counter()
    st = _lib.Stream(pw, 3, _lib.MS_STREAM_PACKED, 2)
    t_created = time.perf_counter()
    got, t_first, sub_done = 0, None, None
    stamps = []
    for b in batches:
        while st.in_flight >= st.capacity:
            r = st.next(); stamps.append(time.perf_counter()); r.close(); got += 1
        st.submit(*b)
    sub_done = time.perf_counter()
    while st.in_flight:
        r = st.next(); stamps.append(time.perf_counter()); r.close(); got += 1
    t_last = time.perf_counter()
    stats = st.stats()
    st.close()
    t_end = time.perf_counter()
    ms = lambda a, b: round((b - a) * 1e3, 2)
    print(f"pass {rep}: total {ms(t0, t_end)} ms | stream created {ms(t0, t_created)} | all submitted at {ms(t0, sub_done)} | first result at {ms(t0, stamps[0])} | "
          f"last two results at {ms(t0, stamps[-2])}, {ms(t0, stamps[-1])} | close {ms(t_last, t_end)} | scan stage work {stats['scan']['ms_work']:.1f} wait_in {stats['scan']['ms_wait_in']:.1f} | "
          f"upload work {stats['upload']['ms_work']:.1f} | copy_out work {stats['copy_out']['ms_work']:.1f} wait_in {stats['copy_out']['ms_wait_in']:.1f}", flush=True)
