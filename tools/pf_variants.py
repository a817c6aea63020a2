#!/usr/bin/env python3
"""Stage times of the scan on one workload, with and without candidate emission (MS_PF_NOEMIT), in ONE process.
Usage (GPU box): python tools/pf_variants.py [workload] [strand] [p-value key]   e.g. c4shard 3 1e-4
Optional environment: N_PWMS (first N motifs only), MAX_W (only motifs of width <= MAX_W), EXTRA_W=33,40 (append one random motif
of each of these widths with a cutoff at about the same hit density)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motifscan_amd import _lib, synth

wl_name = sys.argv[1] if len(sys.argv) > 1 else "c4shard"
strand = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pkey = sys.argv[3] if len(sys.argv) > 3 else "1e-4"
_lib.set_device(0)
wl = synth.workload(wl_name)
vals, widths, cutoffs = synth.load_motif_set(len(wl["widths"]), pkey)
n_pw = int(os.environ.get("N_PWMS", "0")) or len(widths)
max_w = int(os.environ.get("MAX_W", "0"))
mats = synth.matrices_of(vals, widths)
keep = [i for i in range(n_pw) if not max_w or widths[i] <= max_w]
mats = [mats[i] for i in keep]
cuts = [float(cutoffs[i]) for i in keep]
for w in [int(x) for x in os.environ.get("EXTRA_W", "").split(",") if x]:
    m, c = synth.random_motif(w, seed=900 + w, p_value=float(pkey))
    mats.append(m)
    cuts.append(c)
pw = _lib.PwmSet.from_matrices(mats, cuts)
sq = _lib.SeqSet(*wl["sets"][0])
print(f"{wl_name}: {len(mats)} motifs (widths {min(m.shape[1] for m in mats)}..{max(m.shape[1] for m in mats)}), strand mask {strand}, p {pkey}, "
      f"{sq.n_bases / 1e6:.1f} Mbase", flush=True)
os.environ["MS_MEASURE"] = "1"          # opt in to the library's measurement switches
os.environ["MS_PF_CLOCK"] = "1"
for rep in range(3):
    for noemit in (0, 1):
        os.environ["MS_PF_NOEMIT"] = str(noemit)
        rows = []
        for _ in range(5):
            r = _lib.scan(pw, sq, strand)
            st = r.stats()
            rows.append((st["ms_prefilter"], st["ms_exact"], st["ms_sort"], st["ms_finalize"], st["ms_total"]))
            r.close()
        best = min(rows)
        print(f"rep {rep} noemit {noemit}: prefilter {best[0]:.3f} fp64 {best[1]:.3f} sort {best[2]:.3f} finalize {best[3]:.3f} total {best[4]:.3f} ms "
              f"(prefilter all {['%.2f' % x[0] for x in rows]}) tiles {st['n_tiles']} exact motifs {st['n_pwms_exact']} clock {st['pf_clock_mhz']:.0f} MHz "
              f"cand {st['n_candidates']} hits {st['n_hits']}", flush=True)
