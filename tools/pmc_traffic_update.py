#!/usr/bin/env python3
"""profiles/pmc_traffic.json["<workload>_current"] from an evidence pass's PMC summaries (tools/pmc_summary.py output of the separate
`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes): HBM bytes per launch of the pre-filter = (2 x FETCH_SIZE + WRITE_SIZE) KB, FETCH_SIZE
doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 -- stamped with the hash of the kernel source it was measured on, which
bench.py compares before it prints the figure as `roofline.traffic`.
    python tools/pmc_traffic_update.py <workload> <pmc_fetch.csv> <pmc_write.csv> <note>"""
import csv, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workload, f_fetch, f_write, note = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
KERNEL = "prefilter_f6_kernel"


def mean(path, counter):
    for r in csv.DictReader(open(path)):
        if KERNEL in r["kernel"] and r["counter"] == counter:
            return r["kernel"], float(r["mean_per_launch"]), int(r["launches"])
    raise SystemExit(f"{path}: no {counter} row for {KERNEL}")


k, fetch_kb, n_f = mean(f_fetch, "FETCH_SIZE")
_, write_kb, n_w = mean(f_write, "WRITE_SIZE")
sha = hashlib.sha256(open(os.path.join(ROOT, "motifscan_amd", "csrc", "ms_kernels.hip"), "rb").read()).hexdigest()[:16]
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
d = json.load(open(path))
d[workload + "_current"] = {"kernel": k, "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024), "FETCH_SIZE_KB_raw": round(fetch_kb, 1),
                            "WRITE_SIZE_KB_raw": round(write_kb, 1), "launches": [n_f, n_w], "kernel_source_sha16": sha,
                            "from": [os.path.relpath(f_fetch, ROOT), os.path.relpath(f_write, ROOT)], "note": note}
json.dump(d, open(path, "w"), indent=1)
print(workload + "_current", d[workload + "_current"])
