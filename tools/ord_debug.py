import os, sys
sys.path.insert(0, "/root/repo")
os.environ["MS_MEASURE"]="1"; os.environ["MS_ORD_DEBUG"]="1"
from motifscan_amd import _lib, synth
_lib.set_device(0)
wl = synth.workload("c4shard")
vals, widths, cutoffs = synth.load_motif_set(len(wl["widths"]), "1e-4")
pw = _lib.PwmSet(vals, widths, cutoffs)
sq = _lib.SeqSet(*wl["sets"][0])
for i in range(3):
    r = _lib.scan(pw, sq, 3); st = r.stats(); print(i, st["n_passes"], st["n_hits"], "pf %.3f fp64 %.3f sort %.3f fin %.3f total %.3f" % (st["ms_prefilter"], st["ms_exact"], st["ms_sort"], st["ms_finalize"], st["ms_total"])); r.close()
