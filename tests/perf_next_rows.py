#!/usr/bin/env python3
"""Measurements for the SURVEY.md 8(f) rows built on top of the scan path (N1-N3), with the CPU
reference beside them where the reference has a counterpart.  Prints one JSON object.
Usage (GPU box): python tests/perf_next_rows.py   (lives under tests/ because it times the oracle / the real reference beside the GPU rows)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motifscan_amd import _lib, synth, build
from oracle import oracle

_lib.set_device(0)
out = {"device": _lib.device_name(), "host_cores": os.cpu_count()}
vals, widths, cutoffs = synth.load_motif_set(579)
mats = synth.matrices_of(vals, widths)
pw = _lib.PwmSet(vals, widths, cutoffs)

# ---- N1: cutoff builder, the reference's default sampling size (10^6 k-mers of the widest motif's length)
n_k, wmax = 1_000_000, int(widths.max())
kmers, koff = synth.make_regions(n_k, wmax, seed=77, frac_n=0.0, frac_lower=0.0)
sq = _lib.SeqSet(kmers, koff)
ranks = list(build.cutoff_ranks(n_k).values())
_lib.score_ranks(pw, sq, ranks, 3)                       # warm-up
t0 = time.perf_counter(); cut = _lib.score_ranks(pw, sq, ranks, 3); t = time.perf_counter() - t0
ref = oracle.load_reference_ext()
n_cpu = 20000
sub = [kmers[i * wmax:(i + 1) * wmax].tobytes().decode() for i in range(n_cpu)]
ml = [m.tolist() for m in mats]
threads = min(os.cpu_count() or 1, 579)
if ref is not None:
    t0 = time.perf_counter(); sc = ref.c_score(ml, sub, 3, threads); tc = time.perf_counter() - t0
else:
    t0 = time.perf_counter(); sc = oracle.c_score(ml, sub, 3, threads); tc = time.perf_counter() - t0
# parity of the device cutoffs on the sample the CPU scored: same ranks on the same subset
sqs = _lib.SeqSet(kmers[:n_cpu * wmax], koff[:n_cpu + 1])
rk = list(build.cutoff_ranks(n_cpu).values())
dev_small = _lib.score_ranks(pw, sqs, rk, 3)
cpu_small = -np.sort(-np.array(sc), axis=1)[:, rk]
out["N1_cutoff_builder"] = {
    "workload": f"{n_k} background {wmax}-mers x 579 PWMs, both strands, ranks {ranks}",
    "gpu_seconds": t, "gpu_kmer_motifs_per_s": n_k * 579 / t,
    "cpu_kind": "reference" if ref is not None else "port", "cpu_threads": threads,
    "cpu_sample": f"{n_cpu} k-mers, {tc:.2f} s (c_score only; the reference then sorts 579 Python lists)",
    "cpu_kmer_motifs_per_s": n_cpu * 579 / tc, "bit_exact_on_sample": bool(np.array_equal(dev_small, cpu_small))}
sq.close(); sqs.close()

# ---- N2 / N3 on the default bench shard
bases, offsets = synth.make_regions(125_000, 500, seed=1)
t0 = time.perf_counter(); g = _lib.ResidentGenome({"chr": bases}); tg = time.perf_counter() - t0
R = 125_000
ci, st = np.zeros(R, dtype=np.int32), np.arange(R, dtype=np.int64) * 500
g.extract(ci, st, st + 500).close()
t0 = time.perf_counter(); sq = g.extract(ci, st, st + 500); te = time.perf_counter() - t0
out["N3_resident_genome"] = {"pack_seconds_62.5Mbp_incl_H2D": tg, "extract_seconds_125k_regions_x_500bp": te,
                             "extract_bases_per_s": R * 500 / te}
res = _lib.scan(pw, sq, 3)
n0 = res.n_hits
t0 = time.perf_counter(); res.dedup(pw); td = time.perf_counter() - t0
t0 = time.perf_counter(); ns, mx = res.site_tables(R); tt = time.perf_counter() - t0
h = _lib.scan(pw, sq, 3).hits()
t0 = time.perf_counter(); keep = _lib.dedup_keep(h["motif_offsets"], widths, h["seq_idx"], h["pos"], h["score"], h["strand"]); th = time.perf_counter() - t0
out["N2_dedup_and_tables"] = {"hits_before": int(n0), "hits_after": int(res.n_hits), "device_dedup_seconds": td,
                              "host_C_dedup_seconds_same_hits": th, "same_result": bool(keep.sum() == res.n_hits),
                              "site_tables_seconds_incl_D2H_of_P_x_R_x_12B": tt, "table_cells": int(ns.size)}
print(json.dumps(out, indent=1))
