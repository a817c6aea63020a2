import os, sys, time
root = sys.argv[1]
sys.path.insert(0, root)
from motifscan_amd import _lib, synth
_lib.set_device(0)
wl = synth.c4_shard(0, 8)
vals, widths, cutoffs = synth.load_motif_set(len(wl["widths"]), "1e-4")
pw = _lib.PwmSet(vals, widths, cutoffs)
sq = [_lib.SeqSet(*s) for s in wl["sets"]]
rows = []
for i in range(10):
    t0 = time.perf_counter()
    st = []
    for s in sq:
        r = _lib.scan(pw, s, 3); st.append(r.stats()); r.close()
    rows.append((time.perf_counter() - t0) * 1e3)
    last = st
print(root, "shard 1/8 step (2 scans) ms:", ["%.2f" % x for x in rows], "stages of the last scan:", {k: round(last[-1][k], 2) for k in ("ms_prefilter", "ms_exact", "ms_sort", "ms_finalize", "ms_total")}, flush=True)
