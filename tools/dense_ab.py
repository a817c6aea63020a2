import os, sys
sys.path.insert(0, "/root/repo")
from motifscan_amd import _lib, synth
pkey = sys.argv[1]; shard = int(sys.argv[2]) if len(sys.argv) > 2 else 1
_lib.set_device(0)
wl = synth.c4_shard(0, shard)
vals, widths, cutoffs = synth.load_motif_set(len(wl["widths"]), pkey)
sq = _lib.SeqSet(*wl["sets"][0])
for dense in ("0", "1", None):
    os.environ["MS_MEASURE"] = "1"
    if dense is None: os.environ.pop("MS_PF_DENSE", None)
    else: os.environ["MS_PF_DENSE"] = dense
    pw = _lib.PwmSet(vals, widths, cutoffs)
    rows = []
    for i in range(6):
        r = _lib.scan(pw, sq, 3); st = r.stats(); r.close()
        rows.append((st["ms_prefilter"], st["ms_exact"], st["ms_sort"], st["ms_finalize"], st["ms_total"], st["pf_engine"], st["n_candidates"], st["n_hits"]))
    b = min(rows[2:])
    print(f"{sq.n_bases/1e6:.0f} Mbase p {pkey} MS_PF_DENSE={dense}: prefilter {b[0]:.2f} fp64 {b[1]:.2f} sort {b[2]:.2f} finalize {b[3]:.2f} total {b[4]:.2f} ms engines {[x[5] for x in rows]} cand {b[6]} hits {b[7]}", flush=True)
