#!/usr/bin/env python3
"""Where does ms_seqset_create_hostpacked spend its time?  The packer alone (one thread, pageable and pinned input), the whole call at
several thread counts (pageable / pinned input), against the device-packed ms_seqset_create.  GPU box.  python tools/hostpack_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motifscan_amd import _lib, synth
_lib.set_device(0)
b, o = synth.make_regions(250_000, 500, seed=3)
pin = _lib.PinnedBuffer(b.size)
pin.array[:] = b
for name, arr in (("pageable", b), ("pinned", pin.array)):
    t = time.perf_counter(); _lib.host_pack(arr, o); dt = time.perf_counter() - t
    print(f"packer alone, one thread, {name} input: {dt * 1e3:.1f} ms for {b.size / 1e6:.0f} Mbase = {b.size / dt / 1e9:.2f} Gbase/s", flush=True)
for name, arr in (("pageable", b), ("pinned", pin.array)):
    for threads in (0, 1, 4, 8, 16):
        best = 1e9
        for _ in range(4):
            t = time.perf_counter(); sq = _lib.SeqSet(arr, o, host_pack_threads=threads); dt = time.perf_counter() - t
            sq.close()
            best = min(best, dt)
        print(f"whole call, {name} input, {'device pack' if threads == 0 else str(threads) + ' host threads'}: {best * 1e3:.1f} ms", flush=True)
