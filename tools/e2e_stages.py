#!/usr/bin/env python3
"""Where a batch-stream pass of the full configs[3] spends its time: the same batches with the copy-out (MS_STREAM_NO_HITS = counts only),
against the resident step.  Usage (GPU box): python tools/e2e_stages.py [batch_regions] [depth]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motifscan_amd import _lib, synth
_lib.set_device(0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
wl = synth.c4_shard(0, 1)
pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], wl["cutoffs"])
pins, cuts = [], []
for bases, offsets in wl["sets"]:
    pin = _lib.PinnedBuffer(bases.size)
    pin.array[:] = bases
    pins.append(pin)
    n = len(offsets) - 1
    for r0 in range(0, n, batch):
        cuts.append((len(pins) - 1, r0, min(n, r0 + batch)))
def split(c, fr):
    k, r0, r1 = c
    pts = [r0 + int((r1 - r0) * f) for f in fr] + [r1]
    return [(k, x, y) for x, y in zip(pts[:-1], pts[1:]) if y > x]
ramp = split(cuts[0], (0.0, 0.25, 0.5)) + cuts[1:-1] + split(cuts[-1], (0.0, 0.5, 0.75))
def mk(cs):
    out = []
    for k, r0, r1 in cs:
        o = wl["sets"][k][1]
        lo, hi = int(o[r0]), int(o[r1])
        out.append((pins[k].array[lo:hi], np.ascontiguousarray(o[r0:r1 + 1] - lo)))
    return out
units = float(wl["units"])
for name, cs, flags, packed in (("ramp, packed copy-out", ramp, 0, True), ("ramp, counts only (no copy-out)", ramp, _lib.MS_STREAM_NO_HITS, False),
                                ("equal batches, packed copy-out", cuts, 0, True)):
    bs = mk(cs)
    best = None
    for rep in range(4):
        st = {}
        t0 = time.perf_counter()
        dev = {"ms_prefilter": 0.0, "ms_exact": 0.0, "ms_sort": 0.0, "ms_finalize": 0.0, "ms_total": 0.0}
        for res in _lib.scan_stream(pw, iter(bs), 3, flags, depth=depth, packed=packed, stage_stats=st):
            s_ = res.stats()
            for k_ in dev:
                dev[k_] += s_[k_]
            res.close()
        t = time.perf_counter() - t0
        if rep and (best is None or t < best[0]):
            best = (t, st, dev)
    t, st, dev = best
    print(f"{name:34s}: {t * 1e3:7.2f} ms per pass = {units / t:.3e} U/s; {len(bs)} batches; device {({k: round(v, 1) for k, v in dev.items()})}")
    print("      ", {k: {f: round(x, 1) for f, x in v.items()} for k, v in st.items() if isinstance(v, dict)})
