#!/usr/bin/env python3
"""
tools/api_time_reference.py -- BUILD CONTAINER ONLY (reads /root/reference): the reference's own
`Scanner(genome, regions).scan_motifs(pwms)` (scanner.py:44-132 over the real cscore.c built into oracle/_ref) timed on
BASELINE configs[1] (10k x 500 bp x 50 PWMs), with its consumers' access pattern -- the number tools/api_time.py's GPU-side
figures stand beside.  configs[2] is out of its reach in this container (57.9M region lists: SURVEY.md H4), so only the cost of
its empty result shape is extrapolated from 8 motifs.

    python tools/api_time_reference.py [--threads 8] > profiles/archive/r04_api_time_reference.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    import make_golden                                        # import_reference(): the real package + oracle/_ref
    import api_time
    from motifscan_amd import synth
    R = make_golden.import_reference()
    wl = synth.workload("c2")
    bases, offsets = wl["sets"][0]
    text = bases.tobytes().decode()
    regions = [api_time.Region("chr", int(offsets[i]), int(offsets[i + 1])) for i in range(wl["n_regions"])]
    pwms = [api_time.Pwm(m, c, "1e-4") for m, c in zip(synth.matrices_of(wl["pwm_values"], wl["widths"]), wl["cutoffs"])]
    genome = api_time.HostGenome(text)
    out = {"what": "reference Scanner.scan_motifs, BASELINE configs[1]", "reference_version": R["version"], "threads": a.threads,
           "host_cores": os.cpu_count()}
    t0 = time.perf_counter()
    sc = R["scanner"].Scanner(genome=genome, regions=regions, window_size=0, strand="both", p_value="1e-4", remove_dup=True,
                              n_threads=a.threads)
    out["scanner_ctor_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    ms = sc.scan_motifs(pwms)
    out["scan_motifs_s"] = time.perf_counter() - t0
    out["value_api"] = wl["units"] / out["scan_motifs_s"]
    out["n_sites"] = sum(len(x) for per in ms for x in per)
    ids = list(range(2000))
    t0 = time.perf_counter()
    api_time.writer_pattern(ms, ids)
    t = time.perf_counter() - t0
    out["writer_pattern"] = {"regions": len(ids), "seconds": t, "ns_per_motif_region": t / (len(ids) * len(pwms)) * 1e9}
    t0 = time.perf_counter()
    api_time.stats_pattern(ms)
    out["stats_pattern_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    _ = [[[] for _ in range(100_000)] for _ in range(8)]
    out["configs2_empty_result_shape_extrapolated_s"] = (time.perf_counter() - t0) * 579 / 8
    print(json.dumps(out))


if __name__ == "__main__":
    main()
