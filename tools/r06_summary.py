#!/usr/bin/env python3
"""Key numbers of an evidence directory (tools/evidence_round.sh), for DESIGN.md section 7 / profiles/INDEX.md.  python tools/r06_summary.py <dir> [prefix]"""
import csv, glob, json, os, sys
d = sys.argv[1]
def line(f):
    p = os.path.join(d, f)
    if not os.path.exists(p):
        return None
    rows = [l for l in open(p) if l.startswith('{"metric"')]
    return json.loads(rows[-1]) if rows else None
def show(f, extra=()):
    j = line(f)
    if not j:
        print(f, "-- missing"); return
    r = j.get("roofline") or {}; s = j.get("stage_ms_per_scan", {})
    if not r:
        print(f"{f:44s} value {j['value']:.4e}  step {j['ms_per_step']:.2f} ms  " + json.dumps({k: j[k] for k in j if k.startswith('value_') or k in ('config',)})[:700]); return
    out = f"{f:44s} value {j['value']:.4e}  step {j['ms_per_step']:.2f} ms  kernel_ms {r['kernel_ms']:.2f} frac {r['frac']:.4f} traffic {r.get('traffic')}  stages " + " ".join(f"{k[3:]}={v:.2f}" for k, v in s.items())
    if "value_8d_end_to_end" in j:
        e = j["value_end_to_end"]
        out += f"\n    e2e {j['value_8d_end_to_end']:.4e} cli {j['value_8d_cli_job']:.4e} over resident {json.dumps({k: round(v, 3) for k, v in j['end_to_end_over_resident'].items()})} ms/pass {json.dumps({k: round(v, 1) for k, v in e['ms_per_pass'].items()})}"
        out += f"\n    each pass {json.dumps(e.get('ms_each_pass'))}"
        out += f"\n    stages {json.dumps(e['stage_ms_last_pass'].get('pipelined'))}"
    if "scale_projection" in j:
        out += f"\n    scale_projection {json.dumps({k: round(v, 2) for k, v in j['scale_projection']['ms_per_step'].items()})} bound {json.dumps({k: round(v, 2) for k, v in j['scale_projection']['speedup_bound'].items()})}"
    if "cpu_baseline" in j:
        c = j["cpu_baseline"]; out += f"\n    cpu_baseline {c['value']:.3e} on {c['cores']} threads ({c['kind']}), 1 thread {c.get('value_1thread', 0):.3e}; parity {json.dumps(j.get('parity_sample'))}"
    if "value_api" in j:
        a = j["value_api"]["configs2"]; out += f"\n    api c3: scan_motifs {a['scan_motifs_s']:.3f}s again {a['scan_motifs_again_s']:.3f}s rows {a['rows_first_pass_s']:.2f}s writer {a['writer_ns_per_motif_region']:.0f} ns/cell (ref lists {a['writer_on_reference_lists_ns_per_motif_region']:.0f})"
    for k in extra:
        out += f"\n    {k}: {json.dumps(j.get(k))[:400]}"
    print(out)
for f in ("bench_c4.json", "bench_c4_with_traffic.json", "bench_under_rocprof_c4.json"):
    show(f)
for f in sorted(os.listdir(d)):
    if f.startswith("bench_") and f.endswith(".json") and f not in ("bench_c4.json", "bench_c4_with_traffic.json", "bench_under_rocprof_c4.json"):
        show(f, ("counts_check",) if "ranks" in f else ())
ks = os.path.join(d, "kernel_stats_c4.csv")
if os.path.exists(ks):
    print("kernel stats:")
    for r in csv.DictReader(open(ks)):
        if float(r.get("Percentage", 0) or 0) > 0.3:
            print("   ", r["Name"][:70], "calls", r["Calls"], "avg ms", round(float(r["AverageNs"]) / 1e6, 3), "pct", r["Percentage"])
for f in sorted(glob.glob(os.path.join(d, "pmc_*.csv"))):
    print(os.path.basename(f))
    for r in csv.DictReader(open(f)):
        if "prefilter" in r["kernel"]:
            print("   ", r["counter"], r["mean_per_launch"], "launches", r["launches"])
for f in ("pytest_gpu.log", "fuzz.log", "pf_floor.log"):
    p = os.path.join(d, f)
    if os.path.exists(p):
        print(f, "::", " | ".join(l.strip() for l in open(p).read().strip().splitlines()[-8:])[:1500])
