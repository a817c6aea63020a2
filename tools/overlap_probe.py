#!/usr/bin/env python3
"""Feasibility probe (round 6): could the tail of one scan (fp64 stage, ordering, finalize: HBM-bound, ~5.5 ms per 500 Mbase) run BESIDE the next scan's
pre-filter if the pre-filter left one CU per shader engine free (224 of 256 blocks: +1.0 ms, profiles/r06t)?  One thread loops full-size scans with
MS_PF_MAX_BLOCKS=<blocks>; the main thread streams a 1 GB device copy in a loop on another stream and records its bandwidth.  Reported: the
pre-filter's time with and without the copy load, and the copy's bandwidth with and without the scans.   python tools/overlap_probe.py <blocks>"""
import os, sys, threading, time
blocks = sys.argv[1] if len(sys.argv) > 1 else "224"
os.environ["MS_MEASURE"] = "1"
os.environ["MS_PF_MAX_BLOCKS"] = blocks
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motifscan_amd import _lib, synth
_lib.set_device(0)
wl = synth.c4_shard(0, 1)
pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], wl["cutoffs"])
sq = _lib.SeqSet(*wl["sets"][0])
dev = torch.device("cuda", 0)
x = torch.empty(1 << 30, dtype=torch.uint8, device=dev).random_(0, 255)
y = torch.empty_like(x)
s2 = torch.cuda.Stream(device=dev)

def copies(seconds):
    out = []
    t_end = time.time() + seconds
    with torch.cuda.stream(s2):
        while time.time() < t_end:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s2); y.copy_(x, non_blocking=True); e1.record(s2)
            e1.synchronize()
            out.append(2.0 * x.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    return out

def scans(n, stats):
    for _ in range(n):
        r = _lib.scan(pw, sq, 3); stats.append(r.stats()); r.close()

for _ in range(3):
    _lib.scan(pw, sq, 3).close()
st0 = []; scans(6, st0)
bw0 = copies(0.5)
st1 = []
th = threading.Thread(target=scans, args=(14, st1)); th.start()
bw1 = copies(0.25)
th.join()
pf = lambda st: sorted(s["ms_prefilter"] for s in st)[len(st) // 2]
tot = lambda st: sorted(s["ms_total"] for s in st)[len(st) // 2]
med = lambda v: sorted(v)[len(v) // 2]
print(f"blocks {blocks}: pre-filter alone {pf(st0):.2f} ms (scan total {tot(st0):.2f}); beside the copy loop {pf(st1):.2f} ms (scan total {tot(st1):.2f}); "
      f"copy alone {med(bw0):.2f} TB/s; beside the scans median {med(bw1):.2f} TB/s, slowest {min(bw1):.2f}, fastest {max(bw1):.2f} ({len(bw1)} copies of 1 GB)")
