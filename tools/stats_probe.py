import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motifscan_amd import _lib, synth
_lib.set_device(0)
wl = synth.workload("c4shard")
sq = _lib.SeqSet(*wl["sets"][0])
pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], wl["cutoffs"])
for _ in range(8):
    r = _lib.scan(pw, sq, 3); st = r.stats(); r.close()
print({k: st[k] for k in st if k.startswith("n_") or k.startswith("ms_")})
