#!/usr/bin/env python3
"""Is the pre-filter power-limited?  Runs full-size configs[3] scans back to back (the product path) for a few seconds while a thread samples the
board's power and shader clock (hwmon / pp_dpm_sclk / rocm-smi, whatever the box lets an ordinary user read); then the same for an idle device and
for a kernel-free wait.  Usage (GPU box): python tools/power_probe.py [seconds]"""
import glob, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motifscan_amd import _lib, synth

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0


def sources():
    out = {}
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("power1_average", "power1_input", "power1_cap", "power1_cap_max", "freq1_input", "freq2_input", "temp1_input", "temp2_input"):
            p = os.path.join(d, name)
            if os.path.exists(p):
                out.setdefault(d, []).append(name)
    return out


def read(p):
    try:
        return open(p).read().strip()
    except Exception as e:      # noqa: BLE001
        return f"<{type(e).__name__}>"


src = sources()
print("hwmon:", {k: v for k, v in src.items()}, flush=True)
dev = None
for d in src:                   # the device the library uses is card of index 0 on a 1-GPU box: take the one whose power changes under load (all are printed anyway)
    dev = d if dev is None else dev
samples = []
stop = threading.Event()


def sampler():
    while not stop.is_set():
        row = {"t": time.time()}
        for d, names in src.items():
            for n in names:
                if n.startswith("power1_a") or n.startswith("power1_i") or n.startswith("freq1"):
                    row[os.path.basename(d) + "/" + n] = read(os.path.join(d, n))
        samples.append(row)
        time.sleep(0.02)


def summarise(what, t0, t1):
    rows = [r for r in samples if t0 <= r["t"] <= t1]
    keys = sorted(k for k in (rows[0] if rows else {}) if k != "t")
    for k in keys:
        vals = []
        for r in rows:
            try:
                vals.append(float(r[k]))
            except ValueError:
                pass
        if vals:
            vals.sort()
            unit = 1e6 if "power" in k else 1e6
            print(f"  {what:34s} {k:28s} n {len(vals):4d}  median {vals[len(vals) // 2] / unit:8.1f}  p10 {vals[len(vals) // 10] / unit:8.1f}  p90 {vals[len(vals) * 9 // 10] / unit:8.1f}  ({'W' if 'power' in k else 'MHz'})", flush=True)


for d, names in src.items():
    for n in names:
        if "cap" in n:
            print(f"  {os.path.basename(d)}/{n} = {read(os.path.join(d, n))}", flush=True)
_lib.set_device(0)
wl = synth.c4_shard(0, 1)
vals_, widths, cutoffs = synth.load_motif_set(len(wl["widths"]), "1e-4")
pw = _lib.PwmSet(vals_, widths, cutoffs)
sq = _lib.SeqSet(*wl["sets"][0])
for _ in range(3):
    _lib.scan(pw, sq, 3).close()
th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(1.0)
t_idle0, t_idle1 = time.time() - 1.0, time.time()
t0 = time.time()
n = 0
pf = []
while time.time() - t0 < secs:
    r = _lib.scan(pw, sq, 3)
    pf.append(r.stats()["ms_prefilter"])
    r.close()
    n += 1
t1 = time.time()
time.sleep(1.0)
stop.set()
th.join()
pf.sort()
print(f"{n} scans in {t1 - t0:.2f} s; pre-filter median {pf[len(pf) // 2]:.2f} ms (min {pf[0]:.2f}); the pre-filter is {sum(pf) / 1e3 / (t1 - t0):.0%} of the loop's wall time", flush=True)
summarise("idle (1 s before the loop)", t_idle0, t_idle1)
summarise("scan loop (all stages)", t0 + 0.5, t1)
try:
    print(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True, timeout=20).stdout[-1500:])
except Exception as e:      # noqa: BLE001
    print("rocm-smi:", e)
