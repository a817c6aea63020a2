import cProfile, pstats, sys, os, time, io
sys.path.insert(0, "/root/repo")
sys.argv=["x"]
import tools.api_time as A
from motifscan_amd import _lib, scanner, synth
import numpy as np
wl = synth.workload("c3")
bases, offsets = wl["sets"][0]
text = bases.tobytes().decode()
regions = [A.Region("chr", int(offsets[i]), int(offsets[i + 1])) for i in range(wl["n_regions"])]
pwms = [A.Pwm(m, c, "1e-4") for m, c in zip(synth.matrices_of(wl["pwm_values"], wl["widths"]), wl["cutoffs"])]
_lib.set_device(0)
# warm the device context with a tiny scan
w2 = synth.workload("c2")
sc0 = scanner.Scanner(A.HostGenome(w2["sets"][0][0].tobytes().decode()), [A.Region("chr", int(w2["sets"][0][1][i]), int(w2["sets"][0][1][i+1])) for i in range(100)], p_value="1e-4")
sc0.scan_motifs([A.Pwm(m, c, "1e-4") for m, c in zip(synth.matrices_of(w2["pwm_values"], w2["widths"]), w2["cutoffs"])][:5])
genome = A.HostGenome(text)
t0=time.perf_counter()
sc = scanner.Scanner(genome, regions, p_value="1e-4")
t1=time.perf_counter()
pr = cProfile.Profile(); pr.enable()
res = sc.scan_motifs(pwms)
pr.disable()
t2=time.perf_counter()
print("ctor %.4f scan_motifs %.4f"%(t1-t0,t2-t1))
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("cumulative").print_stats(25); print(s.getvalue()[:4000])
