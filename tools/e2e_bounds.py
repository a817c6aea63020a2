#!/usr/bin/env python3
"""What the two stages AROUND the scan can give an end-to-end pass at most: the bench's own configs[3] batches
through the batch stream (ms_stream_*) with the upload stage emptied (the regions cut on the device out of a genome resident in HBM:
nothing crosses the link on the way in, no pack kernel runs beside the pre-filter) and / or the copy-out stage emptied (MS_STREAM_NO_HITS:
nothing crosses the link on the way out) -- upper bounds for host-side 2-bit packing (0.375 instead of 1 byte per base) and for 12-byte
compact hits (12 instead of 16 bytes per hit).  Wall time per pass, best of the last three of five, and the stages' own clocks.
    python tools/e2e_bounds.py          (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motifscan_amd import _lib, synth, dist as msdist

_lib.set_device(0)
wl = synth.c4_shard(0, 1)
pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], wl["cutoffs"])
L = wl["length"]
pins, host_batches, cut_batches = [], [], []
genome = _lib.ResidentGenome({f"set{k}": b for k, (b, o) in enumerate(wl["sets"])})
n_sets = len(wl["sets"])
for k, (bases, offsets) in enumerate(wl["sets"]):
    pin = _lib.PinnedBuffer(max(bases.size, 1)); pin.array[:bases.size] = bases; pins.append(pin)
    for r0, r1 in msdist.batch_bounds(len(offsets) - 1, 125000, ramp=True, max_batch=250000, ramp_up=k == 0, ramp_down=k == n_sets - 1):
        lo, hi = int(offsets[r0]), int(offsets[r1])
        host_batches.append((pin.array[lo:hi], np.ascontiguousarray(offsets[r0:r1 + 1] - lo)))
        cut_batches.append((genome, np.full(r1 - r0, k, dtype=np.int32), offsets[r0:r1].copy(), offsets[r0 + 1:r1 + 1].copy()))
units = float(wl["units"])


def one(batches, flags, packed):
    best = None
    for rep in range(5):
        st = {}
        t0 = time.perf_counter()
        n = 0
        for res in _lib.scan_stream(pw, iter(batches), 3, flags, depth=2, packed=packed, stage_stats=st):
            n += res.n_hits
            res.close()
        t = time.perf_counter() - t0
        if rep >= 2 and (best is None or t < best[0]):
            best = (t, st, n)
    return best


for name, batches, flags, packed in (("host ASCII in, 16-byte hits out (the bench's pipelined leg)", host_batches, 0, True),
                                     ("host ASCII in, NO hits out (counts only)", host_batches, _lib.MS_STREAM_NO_HITS, False),
                                     ("regions cut from HBM, 16-byte hits out", cut_batches, 0, True),
                                     ("regions cut from HBM, NO hits out", cut_batches, _lib.MS_STREAM_NO_HITS, False)):
    t, st, n = one(batches, flags, packed)
    print(f"{name:62s}: {t * 1e3:6.1f} ms per pass = {units / t:.3e} U/s | work ms: upload {st['upload']['ms_work']:.1f} scan {st['scan']['ms_work']:.1f} "
          f"copy-out {st['copy_out']['ms_work']:.1f} | scan waits in {st['scan']['ms_wait_in']:.1f} out {st['scan']['ms_wait_out']:.1f}", flush=True)
# the resident step beside it: two scans of the two sets, hits left in HBM
sqs = [_lib.SeqSet(b, o, keep_ascii=True) for b, o in wl["sets"]]
ts = []
for rep in range(6):
    t0 = time.perf_counter()
    for sq in sqs:
        sq.repack()
        _lib.scan(pw, sq, 3).close()
    ts.append(time.perf_counter() - t0)
print(f"{'device-resident step (repack + scan of both sets, hits stay in HBM)':62s}: {min(ts[2:]) * 1e3:6.1f} ms per step = {units / min(ts[2:]):.3e} U/s")
