#!/usr/bin/env python3
"""Shader clock held inside the pre-filter kernel (s_memtime vs the 100 MHz s_memrealtime), after
>= 2 s of back-to-back launches on the benchmark data.  Usage: python tools/pf_clock.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motifscan_amd import _lib, synth
_lib.set_device(0)
wl = synth.workload(sys.argv[1] if len(sys.argv) > 1 else "c4shard")
pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], wl["cutoffs"])
sq = _lib.SeqSet(*wl["sets"][0])
t0 = time.time()
while time.time() - t0 < 2.5:
    _lib.scan(pw, sq, 3).close()
os.environ["MS_MEASURE"] = "1"          # opt in to the library's measurement switches
os.environ["MS_PF_CLOCK"] = "1"
for _ in range(5):
    r = _lib.scan(pw, sq, 3); st = r.stats(); r.close()
    lds = st["lds_bytes_read"] / (st["ms_prefilter"] * 1e-3)
    peak = 256 * 256 * st["pf_clock_mhz"] * 1e6
    line = (f"prefilter {st['ms_prefilter']:.3f} ms  clock {st['pf_clock_mhz']:.0f} MHz  LDS {lds/1e12:.1f} TB/s = "
            f"{100*lds/peak:.1f}% of 256 B/clk/CU at that clock ({100*lds/(256*256*2.4e9):.1f}% at 2.4 GHz)")
    if st["pf_engine"] == 1:                        # matrix pipe: 32 cycles per v_mfma_i32_32x32x32_i8 per SIMD (tools/ubench)
        cycles = st["ms_prefilter"] * 1e-3 * st["pf_clock_mhz"] * 1e6
        n_mfma = st["mfma_ops"] / 65536 / 1024
        line += f"  matrix pipe busy {100 * 32 * n_mfma / cycles:.1f}% ({cycles / n_mfma:.1f} cycles per instruction per SIMD)"
    print(line, flush=True)
