#!/usr/bin/env python3
"""
tools/api_time.py -- times the drop-in WHERE north_star puts it: `Scanner(genome, regions, ...).scan_motifs(pwms)` and what the
reference's consumers then do with the result (io/__init__.py:23-33 `len(sites[idx])`, `max(site.score ...)`; stats.py:29-31),
at BASELINE configs[1] (10k x 500 bp x 50 PWMs) and configs[2] (100k x 1 kb x 579 PWMs).  Prints one JSON object; bench.py
embeds the same measurement as its `value_api` key.  GPU box only (the product has no CPU path).

    python tools/api_time.py [--configs c2,c3] [--writer-regions 2000]
"""
import argparse
import gc
import json
import os
import resource
import sys
import time
import tracemalloc

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Region:
    __slots__ = ("chrom", "start", "end", "summit")

    def __init__(self, chrom, start, end):
        self.chrom, self.start, self.end, self.summit = chrom, start, end, (start + end) // 2


class Pwm:
    def __init__(self, matrix, cutoff, p_value):
        self.matrix, self.cutoffs, self.length = matrix, {p_value: cutoff}, matrix.shape[1]


class HostGenome:
    """What Scanner reads of motifscan.genome.Genome (scanner.py:81-87): one chromosome holding the regions back to back."""

    def __init__(self, text):
        self._t = text
        self.chrom_sizes = {"chr": len(text)}

    def fetch_sequence(self, chrom, start, end):
        return self._t[start:end]


def writer_pattern(motif_sites, region_ids):
    """io/__init__.py:23-33, verbatim access pattern."""
    n = 0
    for idx in region_ids:
        n_sites, scores = [], []
        for sites in motif_sites:
            num = len(sites[idx])
            n_sites.append(num)
            if num == 0:
                scores.append("NA")
            else:
                scores.append(max([site.score for site in sites[idx]]))
        n += sum(n_sites)
    return n


def stats_pattern(motif_sites):
    """stats.py:27-31, verbatim."""
    return [(len(sites), sum([len(sites_by_region) > 0 for sites_by_region in sites])) for sites in motif_sites]


def measure(name, writer_regions=None, p_value="1e-4", resident=False):
    from motifscan_amd import _lib, scanner, synth
    wl = synth.workload(name)
    bases, offsets = wl["sets"][0]
    n_regions, L, P = wl["n_regions"], wl["length"], wl["n_pwms"]
    text = bases.tobytes().decode()
    regions = [Region("chr", int(offsets[i]), int(offsets[i + 1])) for i in range(n_regions)]
    pwms = [Pwm(m, c, p_value) for m, c in zip(synth.matrices_of(wl["pwm_values"], wl["widths"]), wl["cutoffs"])]
    genome = _lib.ResidentGenome({"chr": bases}) if resident else HostGenome(text)
    units = float(wl["units"])
    out = {"workload": name, "n_regions": n_regions, "region_bp": L, "n_pwms": P, "genome": "resident in HBM" if resident else "host strings"}

    # warm the device path (pools, plan) exactly as a second CLI run would find it
    scanner.Scanner(genome, regions[:256], p_value=p_value).scan_motifs(pwms)

    t0 = time.perf_counter()
    sc = scanner.Scanner(genome, regions, window_size=0, strand="both", p_value=p_value, remove_dup=True)
    out["scanner_ctor_s"] = time.perf_counter() - t0          # a7: one fetch_sequence per region (host genome), as the reference
    gc.collect()
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    tracemalloc.start()
    t0 = time.perf_counter()
    ms = sc.scan_motifs(pwms)
    out["scan_motifs_s"] = time.perf_counter() - t0
    py_bytes = tracemalloc.get_traced_memory()[0]
    tracemalloc.stop()
    out["scan_motifs_python_heap_bytes"] = int(py_bytes)      # numpy bookkeeping + the view; the hit arrays are pinned library memory
    out["n_sites"] = ms.n_sites
    out["flat_arrays_bytes"] = int(ms.n_sites * 25 + 8 * (P + 1))
    out["max_rss_growth_bytes"] = int((resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - rss0) * 1024)
    out["value_api"] = units / out["scan_motifs_s"]
    out["value_api_with_ctor"] = units / (out["scan_motifs_s"] + out["scanner_ctor_s"])
    # second call: the steady state of a process that scans input then control regions (cli/scan.py:76-86)
    t0 = time.perf_counter()
    ms2 = sc.scan_motifs(pwms)
    out["scan_motifs_again_s"] = time.perf_counter() - t0
    del ms2

    # the reference consumers' own loops over the result, MEASURED over everything (round 4 extrapolated from a sample): iteration hands
    # out one real list per motif (sites.RegionRow), built on the first pass
    t0 = time.perf_counter()
    writer_pattern(ms, [0])                                    # the first region's inner loop walks every motif: every row gets built
    out["rows_first_pass_s"] = time.perf_counter() - t0
    ids = list(range(n_regions if writer_regions is None else min(writer_regions, n_regions)))
    t0 = time.perf_counter()
    n_seen = writer_pattern(ms, ids)
    t = time.perf_counter() - t0
    out["writer_pattern"] = {"regions": len(ids), "seconds": t, "ns_per_motif_region": t / (len(ids) * P) * 1e9, "sites_seen": n_seen,
                             "all_regions_s": t * n_regions / len(ids), "measured_over_all_regions": len(ids) == n_regions}
    t0 = time.perf_counter()
    st = stats_pattern(ms)
    t = time.perf_counter() - t0
    assert [b for _, b in st] == ms.n_regions_with_site.tolist()
    out["stats_pattern"] = {"motifs": P, "seconds": t, "all_motifs_s": t}
    # the floor: the same writer loop over the REFERENCE's own structure (real nested lists of lists), for a slice of the motifs
    m_take = min(P, 40)
    eager = [[list(x) for x in row] for row in ms[:m_take]]
    sample = ids[:min(len(ids), 20_000)]
    t0 = time.perf_counter()
    writer_pattern(eager, sample)
    out["writer_pattern_on_reference_lists_ns_per_motif_region"] = (time.perf_counter() - t0) / (len(sample) * m_take) * 1e9
    del eager
    t0 = time.perf_counter()
    ns, mx = ms.site_counts(), ms.max_scores()
    out["vectorised_tables_s"] = time.perf_counter() - t0      # what formats.write_sites_table takes instead of the double loop
    assert int(ns.sum()) == ms.n_sites and np.array_equal((ns > 0).sum(axis=1), ms.n_regions_with_site)
    if n_regions * P <= 1_000_000:                             # the reference's eager shape, where it is affordable at all
        t0 = time.perf_counter()
        eager = ms.to_lists()
        out["to_lists_s"] = time.perf_counter() - t0
        assert ms == eager
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c2,c3")
    ap.add_argument("--writer-regions", type=int, default=None, help="default: every region (measured, not extrapolated)")
    ap.add_argument("--resident", action="store_true", help="also time a ResidentGenome (regions cut on the device, no fetch_sequence)")
    a = ap.parse_args()
    from motifscan_amd import _lib
    if _lib.device_count() < 1:
        raise SystemExit("api_time.py needs an MI355X; there is no CPU fallback")
    _lib.set_device(0)
    res = {"device": _lib.device_name(), "host_cores": os.cpu_count(), "runs": []}
    for name in a.configs.split(","):
        res["runs"].append(measure(name, a.writer_regions))
        if a.resident:
            res["runs"].append(measure(name, a.writer_regions, resident=True))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
