#!/usr/bin/env python3
"""Wall time of one ms_scan call at configs[1] size (10k x 500 bp x 50 PWMs), through the C-ABI, against its device stages."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motifscan_amd import _lib, synth
_lib.set_device(0)
wl = synth.workload("c2")
pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], wl["cutoffs"])
sq = _lib.SeqSet(*wl["sets"][0], keep_ascii=True)
L = _lib.lib()
for mode in ("predicted", "two-sync"):
    if mode == "two-sync":
        os.environ["MS_MEASURE"] = "1"; os.environ["MS_NO_PREDICT"] = "1"
    for _ in range(50):
        _lib.scan(pw, sq, 3).close()
    N = 400
    t0 = time.perf_counter()
    for _ in range(N):
        h = ctypes.c_void_p()
        L.ms_scan(pw.h, sq.h, 3, 0, ctypes.byref(h))
        L.ms_result_free(h)
    t_raw = (time.perf_counter() - t0) / N
    t0 = time.perf_counter()
    for _ in range(N):
        r = _lib.scan(pw, sq, 3)
        r.close()
    t_py = (time.perf_counter() - t0) / N
    t0 = time.perf_counter()
    for _ in range(N):
        sq.repack()
        r = _lib.scan(pw, sq, 3)
        r.close()
    t_rp = (time.perf_counter() - t0) / N
    st = _lib.scan(pw, sq, 3).stats()
    print(f"{mode:10s}: ms_scan + free through ctypes {t_raw * 1e3:.3f} ms; _lib.scan + close {t_py * 1e3:.3f} ms; with repack {t_rp * 1e3:.3f} ms; "
          f"device stages {st['ms_total']:.3f} ms (prefilter {st['ms_prefilter']:.3f} fp64 {st['ms_exact']:.3f} sort {st['ms_sort']:.3f} finalize {st['ms_finalize']:.3f}) "
          f"passes {st['n_passes']} hits {st['n_hits']}", flush=True)
