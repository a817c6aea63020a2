import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from motifscan_amd import _lib, synth
_lib.set_device(0)
wl = synth.workload("c2")
pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], wl["cutoffs"])
bases, offs = wl["sets"][0]
bases = np.frombuffer(bases, dtype=np.uint8) if not isinstance(bases, np.ndarray) else bases
print("regions", len(offs) - 1, "bases", len(bases), "P", len(wl["widths"]), "maxW", wl["widths"].max())
for mult in (0.1, 0.25, 0.5, 1, 2, 4, 8):
    if mult <= 1:
        R = int((len(offs) - 1) * mult)
        b, o = bases[:offs[R]], offs[:R + 1]
    else:
        m = int(mult)
        b = np.tile(bases, m)
        o = np.concatenate([offs[:-1] + k * offs[-1] for k in range(m)] + [[offs[-1] * m]])
    sq = _lib.SeqSet(b, o)
    ms = []
    for _ in range(6):
        r = _lib.scan(pw, sq, 3); st = r.stats(); r.close(); ms.append((st["ms_prefilter"], st["ms_exact"], st["ms_sort"], st["ms_total"]))
    best = min(ms)
    print("bases %9d: prefilter %.3f exact %.3f sort %.3f total %.3f ms  hits %d cand %d" % (len(b), *best, st["n_hits"], st["n_candidates"]))
    sq.close()
