import os, sys
sys.path.insert(0, "/root/repo")
from motifscan_amd import _lib, synth
_lib.set_device(0)
wl = synth.c4_shard(0, 1)
vals, widths, cutoffs = synth.load_motif_set(len(wl["widths"]), "1e-4")
pw = _lib.PwmSet(vals, widths, cutoffs)
sq = _lib.SeqSet(*wl["sets"][0])
os.environ["MS_MEASURE"] = "1"; os.environ["MS_PF_CLOCK"] = "1"
ppw = sq.n_bases / 64 / 4096
for mode, what in ((1, "no hand-off"), (6, "no hand-off, paired two-block row tiles without their operand reads"), (3, "reads + matrix only"), (1, "no hand-off (again)")):
    os.environ["MS_PF_NOEMIT"] = str(mode)
    best = None
    for _ in range(6):
        r = _lib.scan(pw, sq, 3); st = r.stats(); r.close()
        if best is None or st["ms_prefilter"] < best[0]: best = (st["ms_prefilter"], st["pf_clock_mhz"])
    print(f"mode {mode} ({what}): prefilter {best[0]:.3f} ms at {best[1]:.0f} MHz = {best[0]*1e-3*best[1]*1e6/ppw:.0f} cycles per wave and pass", flush=True)
