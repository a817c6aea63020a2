#!/usr/bin/env python3
"""The pre-filter's floor table: the PRODUCT kernel and its four compile-time cuts (ms_kernels.hip, FLOOR: no run-time switch
inside, what is left runs exactly as in the product) on one full-size configs[3] set, same box, interleaved rounds, with the shader clock
sampled from hwmon while each variant loops -- and the issue model beside them:

    matrix pipe   = N_mfma x 32 clk / (1024 SIMDs x f)                         (one 32x32x64 fp6 x fp4 instruction occupies a SIMD's pipe 32 cycles)
    issue floor   = max(matrix pipe, N_valu x c_valu / (1024 x f))             (VALU and matrix instructions of DIFFERENT waves issue side by side)

Usage (GPU box): python tools/pf_floor.py [p-value key] [rounds]"""
import glob, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pkey = sys.argv[1] if len(sys.argv) > 1 else "1e-4"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
CHILD = r'''
import os, sys, json, time, glob, threading
sys.path.insert(0, %r)
from motifscan_amd import _lib, synth
_lib.set_device(0)
wl = synth.c4_shard(0, 1)
vals, widths, cutoffs = synth.load_motif_set(len(wl["widths"]), %r)
pw = _lib.PwmSet(vals, widths, cutoffs)
sq = _lib.SeqSet(*wl["sets"][0])
freq = [p for p in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")]
power = [p for p in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")] + [p for p in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")]
samples, stop = [], threading.Event()
def sampler():
    while not stop.is_set():
        try:
            samples.append((float(open(freq[0]).read()) / 1e6 if freq else 0.0, float(open(power[0]).read()) / 1e6 if power else 0.0))
        except Exception:
            pass
        time.sleep(0.01)
for _ in range(3):
    _lib.scan(pw, sq, 3).close()
th = threading.Thread(target=sampler); th.start()
rows = []
t0 = time.time()
while time.time() - t0 < 1.5 or len(rows) < 8:
    r = _lib.scan(pw, sq, 3); st = r.stats(); rows.append(st["ms_prefilter"]); last = st; r.close()
stop.set(); th.join()
rows.sort()
mhz = sorted(x[0] for x in samples if x[0] > 0); watts = sorted(x[1] for x in samples if x[1] > 0)
print(json.dumps({"ms_median": rows[len(rows) // 2], "ms_min": rows[0], "n": len(rows), "mhz_median": mhz[len(mhz) // 2] if mhz else 0.0,
                  "watts_median": watts[len(watts) // 2] if watts else 0.0, "mfma_ops": last["mfma_ops"], "mfma_ops_algorithmic": last["mfma_ops_algorithmic"], "n_bases": last["n_bases"]}))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), pkey)
NAMES = {0: "product kernel", 1: "cut 1: inspection, no hand-off", 2: "cut 2: operand reads + matrix instructions", 3: "cut 3: matrix instructions alone", 4: "cut 4: set-up only"}
import json
res = {k: [] for k in NAMES}
for rd in range(rounds):
    for k in NAMES:
        env = dict(os.environ)
        if k:
            env.update(MS_MEASURE="1", MS_PF_FLOOR=str(k))
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("variant", k, "failed:", out.stderr[-400:], flush=True); continue
        res[k].append(json.loads(line[-1]))
print(f"# pre-filter floor table, one 500-Mbase set x 579 PWMs, both strands, p = {pkey}; {rounds} interleaved rounds (median-of-loop per round)")
base = None
for k, name in NAMES.items():
    if not res[k]:
        continue
    ms = sorted(r["ms_median"] for r in res[k]); mhz = sorted(r["mhz_median"] for r in res[k]); w = sorted(r["watts_median"] for r in res[k])
    m = ms[len(ms) // 2]
    base = base or m
    print(f"{name:46s} {m:7.2f} ms  (rounds: {' '.join('%.2f' % x for x in ms)})   {m / base:5.2f} of the product")
r0 = res[0][0] if res[0] else None
if r0:
    n_mfma = r0["mfma_ops"] / 131072.0
    print(f"# issue model: {n_mfma / 1e6:.1f} M matrix instructions per launch x 32 clk / 1024 SIMDs:")
    for f in (2400.0, 2250.0, 2060.0):          # nominal; what the product kernel holds under the power limit; what an MFMA-only loop holds (profiles/r05_power_probe.log -- hwmon sampled from a
        if f > 0:                                 # 1.5-s loop does not resolve the clock, so it is not printed here)
            print(f"#   matrix pipe 100 % busy at {f:.0f} MHz: {n_mfma * 32 / 1024 / (f * 1e6) * 1e3:.2f} ms")
    print(f"#   useful fraction of the issued matrix work: {r0['mfma_ops_algorithmic'] / r0['mfma_ops']:.3f} (one-hot k-slots x width padding)")
