import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from motifscan_amd import _lib, synth
_lib.set_device(0)
vals, widths, cutoffs = synth.load_motif_set(579)
pw = _lib.PwmSet(vals, widths, cutoffs)
b, o = synth.make_regions(100, 500, seed=1)
for _ in range(3):
    t0 = time.perf_counter()
    st = _lib.Stream(pw, 3, 0, 2)
    t1 = time.perf_counter()
    st.close()
    t2 = time.perf_counter()
    print(f"create {1e3*(t1-t0):.3f} ms close {1e3*(t2-t1):.3f} ms")
for _ in range(3):
    t0 = time.perf_counter()
    n = 0
    for res in _lib.scan_stream(pw, iter([(b, o)]), 3, 0, depth=2, packed=True):
        n += res.n_hits; res.close()
    print(f"one tiny batch through a fresh stream: {1e3*(time.perf_counter()-t0):.3f} ms")
