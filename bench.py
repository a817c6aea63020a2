#!/usr/bin/env python3
"""
bench.py -- the measurement contract of the PWM scan path.

    python bench.py --gpus N --steps K --warmup W [--workload c4shard|c3|c2]
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

A "step" is one pass of the hot path over this rank's synthetic batch, starting from ASCII bases
resident in HBM: for each region set (input, control)  pack (sequence extraction -> 2-bit codes +
N mask)  ->  integer pre-filter  ->  fp64 re-scoring  ->  ordering  ->  coordinates / per-motif
region counts;  then the single all-reduce of the per-motif region counts (N > 1).  Results stay
in HBM (hits are not copied to the host inside the timed region; the PCIe-inclusive figure is in
DESIGN.md).

Metric: scanned bp x motifs per second (BASELINE.json), whole job over all ranks; weak scaling
(every rank scans its own fixed-size shard; at N = 8 the default workload is BASELINE.json
configs[3]: 1M input + 1M control regions x 500 bp x 579 PWMs).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12                 # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
LDS_PEAK = 256 * 256 * 2.4e9      # B/s: 256 B/clk/CU (ds_read_b128) x 256 CUs x 2.4 GHz
I8_MFMA_PEAK = 256 * 4 * 2048 * 2.4e9   # op/s: v_mfma_i32_32x32x32_i8 = 65536 ops per 32 cycles per SIMD (measured, tools/ubench),
                                        # 1024 SIMDs, 2.4 GHz = 5.03e15 = 2 x the dense bf16 peak of MI355X_MICROARCH.md


def cpu_baseline(wl, seconds_target=12.0):
    """The reference's CPU scanner on a bounded sample of the same workload, on this box's host
    cores.  kind = "reference": the real cscore.c (oracle/_ref, built in the build container);
    otherwise kind = "port": the oracle's C restatement."""
    from oracle import oracle
    from motifscan_amd import synth
    cores = os.cpu_count() or 1
    bases, offsets = wl["sets"][0]
    L, P = wl["length"], wl["n_pwms"]
    mats = [m.tolist() for m in synth.matrices_of(wl["pwm_values"], wl["widths"])]
    cuts = wl["cutoffs"].tolist()
    ref = oracle.load_reference_ext()

    def run(n_regions, threads):
        raw = bases[:int(offsets[n_regions])].tobytes()
        if ref is not None:
            seqs = [raw[int(offsets[i]):int(offsets[i + 1])].decode() for i in range(n_regions)]
            t0 = time.perf_counter()
            ref.c_scan_motif(mats, cuts, seqs, 3, threads)
            return time.perf_counter() - t0
        t0 = time.perf_counter()
        oracle.scan_arrays(wl["pwm_values"], wl["widths"], wl["cutoffs"], raw, offsets[:n_regions + 1], 3, threads)
        return time.perf_counter() - t0

    threads = min(cores, P)                        # the reference's work unit is one whole PWM (cscore.c:181-186)
    n_max = len(offsets) - 1
    n, t = min(64, n_max), 0.0
    for _ in range(4):                             # grow the sample until one run is a few seconds, then scale to target
        t = run(n, threads)
        if t >= 3.0 or n >= n_max:
            break
        n = int(min(n_max, max(2 * n, n * 4.0 / max(t, 1e-3))))
    if t < 0.6 * seconds_target and n < n_max:
        n = int(min(n_max, n * seconds_target / max(t, 1e-3)))
        t = run(n, threads)
    units = int(offsets[n]) * P
    out = {"value": units / t, "unit": "bp*motifs/s", "cores": threads,
           "kind": "reference" if ref is not None else "port",
           "sample": f"first {n} regions x {L} bp x {P} PWMs of the same workload, both strands, {t:.1f} s wall",
           "host_cores_total": cores}
    n0 = 16
    t1 = run(max(n // max(threads, 1), n0), 1)
    out["value_1thread"] = int(offsets[max(n // max(threads, 1), n0)]) * P / t1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c4shard", choices=["c4shard", "c3", "c2", "c5shard", "c5regions", "tiny"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch                                   # device memory / streams / torch.distributed only
    import torch.distributed as dist
    from motifscan_amd import _lib, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available() or _lib.device_count() < 1:
        raise RuntimeError("bench.py needs an MI355X; there is no CPU fallback")
    # MS_BENCH_BACKEND=gloo (test aid only): lets the N > 1 code path run on a box with fewer GPUs than ranks
    # (ranks then share devices; RCCL itself refuses two ranks on one GPU)
    backend = os.environ.get("MS_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % _lib.device_count()
    torch.cuda.set_device(dev_index)               # before the process group: RCCL binds to the current device
    _lib.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if world != a.gpus and rank == 0:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    wl = synth.workload("c5shard" if a.workload == "c5regions" else a.workload, rank=rank)
    P = wl["n_pwms"]
    pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], wl["cutoffs"])
    genome = None
    if "genome" in wl:                             # configs[4]: windows cut from a genome that is resident in HBM
        genome = _lib.ResidentGenome({"chr": wl["genome"]})
        seqsets = [None]
    else:
        seqsets = [_lib.SeqSet(b, o, keep_ascii=True) for b, o in wl["sets"]]
    counts = torch.zeros(len(seqsets) * P, dtype=torch.int64, device=dev)

    def step():
        stats = []
        for s, sq in enumerate(seqsets):
            if genome is not None and a.workload == "c5shard":
                # fixed-stride sweep: the span is scanned once, hits are handed to the windows that hold them
                res = _lib.scan_sweep(pw, genome, "chr", 0, len(wl["genome"]), synth.C5_SHARD["window"], synth.C5_SHARD["stride"], 3)
            elif genome is not None:               # the same windows as an explicit region list -> bit-level gather on the device
                sq = genome.extract(*wl["windows"])
                res = _lib.scan(pw, sq, 3)
                sq.close()
            else:
                sq.repack()                        # extraction: resident ASCII -> 2-bit codes + N mask
                res = _lib.scan(pw, sq, 3)
            stats.append(res.stats())
            counts[s * P:(s + 1) * P] = torch.from_numpy(res.region_counts()).to(dev, non_blocking=False)
            res.close()
        if world > 1:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM)      # the path's one collective (stats.py:29-31 input)
        return stats

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    all_stats = []
    for _ in range(a.steps):
        all_stats.extend(step())
    fence()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    units = torch.tensor([float(wl["units"])], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    total_units = float(units.item())

    if rank == 0:
        n_launch = len(all_stats)
        pf_ms = sum(s["ms_prefilter"] for s in all_stats) / n_launch          # HIP events on the library's stream
        alg_bytes = sum(s["hbm_bytes_algorithmic"] for s in all_stats) / n_launch
        lds_bytes = sum(s["lds_bytes_read"] for s in all_stats) / n_launch
        windows = sum(s["n_windows"] for s in all_stats) / n_launch
        achieved = alg_bytes / (pf_ms * 1e-3)
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")             # recorded PMC pass (rocprofv3 --pmc), per launch
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(a.workload, {}).get("hbm_bytes_per_launch")
            except (OSError, ValueError):
                traffic = None
        engine = all_stats[0]["pf_engine"]
        hbm = {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
               "traffic": traffic, "kernel": "prefilter_kernel", "kernel_ms": pf_ms, "algorithmic_bytes_per_launch": alg_bytes}
        if engine >= 1:
            # dominant kernel = prefilter_mfma_kernel, bound by the matrix pipe: algorithmic ops =
            # SURVEY.md 8(d)'s "one add per (window, column, strand)" counted as a multiply-add (2 ops); what the
            # kernel ISSUES is 4x that (one-hot: 4 k-slots per base) plus padding of widths to 8 columns
            alg_ops = sum(s["mfma_ops_algorithmic"] for s in all_stats) / n_launch
            issued = sum(s["mfma_ops"] for s in all_stats) / n_launch
            roofline = {"bound": "mfma", "achieved": alg_ops / (pf_ms * 1e-3) / 1e12, "peak": I8_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                        "frac": alg_ops / (pf_ms * 1e-3) / I8_MFMA_PEAK, "traffic": traffic, "kernel": "prefilter_mfma_kernel",
                        "kernel_ms": pf_ms, "algorithmic_ops_per_launch": alg_ops, "dtype": "int8 x int8 -> int32"}
            on_chip = {"bound": "matrix pipe (issued int8 ops incl. one-hot zeros and width padding)",
                       "achieved": issued / (pf_ms * 1e-3) / 1e12, "peak": I8_MFMA_PEAK / 1e12, "unit": "TOP/s",
                       "frac": issued / (pf_ms * 1e-3) / I8_MFMA_PEAK, "issued_ops_per_launch": issued,
                       "lds_TBps": lds_bytes / (pf_ms * 1e-3) / 1e12, "windows_per_s_kernel": windows / (pf_ms * 1e-3)}
        else:
            roofline = hbm
            # the stream that actually binds the engine-0 kernel: PWM 2-mer tables read from LDS
            on_chip = {"bound": "lds", "achieved": lds_bytes / (pf_ms * 1e-3) / 1e12, "peak": LDS_PEAK / 1e12,
                       "unit": "TB/s", "frac": lds_bytes / (pf_ms * 1e-3) / LDS_PEAK,
                       "windows_per_s_kernel": windows / (pf_ms * 1e-3)}
        line = {
            "metric": "scanned bp*motifs per second (region_bp x n_motifs), both strands, p=1e-4 cutoffs",
            "value": total_units * a.steps / elapsed,
            "unit": "bp*motifs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f64 (every hit decision and score, as in the reference) behind an int8 one-hot matrix-core pre-filter (i32 accumulate)"
                      if engine == 1 else
                      "f64 (every hit decision and score, as in the reference) behind an int8 Walsh-form matrix-core pre-filter (i32 accumulate)"
                      if engine == 2 else
                      "f64 (every hit decision and score, as in the reference) behind a u32 pre-filter of three packed 10-bit fixed-point fields"),
            "data": "synthetic",
            "config": {"workload": {"c4shard": "BASELINE configs[3] per-GPU shard: (125k input + 125k control) regions x 500 bp x 579 PWMs "
                                               "(N=8 is the full 1M+1M config)",
                                    "c3": "BASELINE configs[2]: 100k x 1 kb regions x 579 PWMs",
                                    "c2": "BASELINE configs[1]: 10k x 500 bp regions x 50 PWMs",
                                    "c5shard": "BASELINE configs[4] per-GPU shard: 375 Mbp of genome resident in HBM as 200 bp windows stride 50 "
                                               "(7.5M windows) x 579 PWMs, through ms_scan_sweep (every base scored once)",
                                    "c5regions": "BASELINE configs[4] per-GPU shard: the same 7.5M windows handed over as an explicit region list",
                                    "tiny": "smoke"}[a.workload],
                       "regions_per_gpu": wl["n_regions"] * max(len(wl["sets"]), 1), "region_bp": wl["length"], "n_pwms": P,
                       "strands": "both", "p_value": "1e-4", "sharding": f"regions over {world} GPU(s), 1 all-reduce of int64[{len(seqsets) * P}]"},
            "roofline": roofline,
            "roofline_hbm": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK, "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes},
            "roofline_on_chip": on_chip,
            "stage_ms_per_scan": {k: sum(s[k] for s in all_stats) / n_launch
                                  for k in ("ms_prefilter", "ms_exact", "ms_sort", "ms_finalize", "ms_total")},
            "hits_per_scan": sum(s["n_hits"] for s in all_stats) / n_launch,
            "candidates_per_scan": sum(s["n_candidates"] for s in all_stats) / n_launch,
        }
        if world == 1 and not a.no_cpu_baseline and wl["sets"]:
            line["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
