"""
motifscan_amd.dist -- multi-GPU form of the scan path: one process per GPU, regions sharded,
ONE collective.

Regions are independent units (windows never cross a region, cscore.c:336-340; de-dup is per
(motif, region, strand), scanner.py:176-193), so the path shards with no data-path exchange:
every rank gets a contiguous block of regions (balanced by bases, not by count) and ALL PWMs.
The only cross-rank quantity the downstream statistics need is, per motif, the number of
regions with >= 1 site in the input set and in the control set (stats.py:29-31): a single
all-reduce(sum) of int64[n_sets * n_pwms] (9 KB at 579 motifs; latency-bound, so ring vs tree
and xGMI link bandwidth are irrelevant).  Hit lists stay on the rank that produced them;
`seq_idx + shard_start` is the global region index, so concatenating the ranks' lists in rank
order reproduces the single-GPU order within each motif.

The reference has no counterpart (it has no distributed code at all); the consumer of the
reduced counts is `motifscan.stats.motif_enrichment`, restated below only as far as needed to
show the hand-over (Fisher exact, fold change).
"""
import numpy as np


def shard_bounds(offsets, world_size):
    """Contiguous region blocks [r0, r1) per rank, balanced by number of bases."""
    offsets = np.asarray(offsets, dtype=np.int64)
    n_regions = len(offsets) - 1
    total = int(offsets[-1])
    cuts = [0]
    for k in range(1, world_size):
        target = total * k // world_size
        r = int(np.searchsorted(offsets, target, side="left"))
        cuts.append(min(max(r, cuts[-1]), n_regions))
    cuts.append(n_regions)
    return [(cuts[k], cuts[k + 1]) for k in range(world_size)]


def take_shard(bases, offsets, r0, r1):
    offsets = np.asarray(offsets, dtype=np.int64)
    lo, hi = int(offsets[r0]), int(offsets[r1])
    return bases[lo:hi], offsets[r0:r1 + 1] - lo


def batch_bounds(n, batch_regions, ramp=True, max_batch=None, ramp_up=True, ramp_down=True):
    """[(r0, r1), ...] covering n regions.  A stream's pass begins with an upload and ends with a copy-out that nothing overlaps, and
    every batch in between pays fixed costs (launch tails, a sort's small passes): so the batches GROW from batch_regions / 4 by
    doubling up to max_batch (default 2 x batch_regions) at the start of a pass, stay there, and shrink the same way at its end
    (ramp_up / ramp_down: whether this set begins / ends the pass).  ramp=False: equal batches of batch_regions.
    Round 2 cut only the first and last batch (1/4 + 1/4 + 1/2): 20 batches per configs[3] pass against 14 now, 62.5 against
    65.5 ms (profiles/archive/r03j_bench_m250000.json; batches of 500k regions make the copy-out stage the bottleneck: 67.8 ms)."""
    step = max(1, int(batch_regions))
    if not ramp or n < 4 * step:
        bounds = [(r0, min(n, r0 + step)) for r0 in range(0, max(n, 1), step)]
        if not ramp or len(bounds) < 4:
            return bounds

        def split(b, fr):
            pts = [b[0] + int((b[1] - b[0]) * f) for f in fr] + [b[1]]
            return [(x, y) for x, y in zip(pts[:-1], pts[1:]) if y > x]
        return (split(bounds[0], (0.0, 0.25, 0.5)) if ramp_up else [bounds[0]]) + bounds[1:-1] + (split(bounds[-1], (0.0, 0.5, 0.75)) if ramp_down else [bounds[-1]])
    big = max(step, int(max_batch) if max_batch else 2 * step)
    up, size = [], max(1, step // 4)
    while size < big:                                   # step/4, step/4, step/2, step, 2 step, ... (< big)
        up.append(size)
        if len(up) >= 2:
            size *= 2
    head = up if ramp_up else []
    tail = up[::-1] if ramp_down else []
    while sum(head) + sum(tail) > n and (head or tail):  # a short set: drop the largest ramp pieces
        if len(head) >= len(tail) and head:
            head.pop()
        elif tail:
            tail.pop(0)
    middle = n - sum(head) - sum(tail)
    n_big = max(1, -(-middle // big)) if middle > 0 else 0
    sizes = list(head)
    for i in range(n_big):                              # the middle in equal parts of at most `big`
        sizes.append(middle // n_big + (1 if i < middle % n_big else 0))
    sizes += tail
    bounds, r0 = [], 0
    for sz in sizes:
        if sz > 0:
            bounds.append((r0, r0 + sz))
            r0 += sz
    assert r0 == n
    return bounds


def gpu_scan(pwm_values, widths, cutoffs, bases, offsets, strand, batch_regions=125_000):
    """Local scan on this process's GPU -> (hits dict, region_counts).  The shard goes through an ms_stream in batches of
    batch_regions regions (upload + pack | scan | copy-out overlapped, SURVEY.md 8(e) "Scaling risks"); the batches are
    merged back into the single-call order."""
    from . import _lib
    offsets = np.asarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    pw = _lib.PwmSet(pwm_values, widths, cutoffs)
    bounds = batch_bounds(n, batch_regions)
    counts = np.zeros(len(widths), dtype=np.int64)
    parts = []
    try:
        # (the 12-byte compact copy-out: every batch of at most 2 x batch_regions regions fits it; a batch that does not leaves in the 16-byte form)
        gen = _lib.scan_stream(pw, (take_shard(bases, offsets, r0, r1) for r0, r1 in bounds), strand, packed=12)
        for (r0, _), res in zip(bounds, gen):
            parts.append((res.hits(packed=True), r0))
            counts += res.region_counts()
            res.close()
        return _lib.merge_hits(parts, len(widths)), counts
    finally:
        pw.close()


def allreduce_counts(counts, device=None):
    """Sum int64 counts over all ranks (RCCL when `device` is a cuda device, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(np.ascontiguousarray(counts, dtype=np.int64))
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def scan_sharded(pwm_values, widths, cutoffs, sets, rank, world_size, strand=3, scan_fn=gpu_scan, device=None):
    """Scan this rank's shard of every region set; all-reduce the per-motif region counts.

    sets: list of (bases uint8 array, offsets) -- e.g. [input, control].
    Returns dict(hits=[per set: hits dict with GLOBAL seq_idx], counts=int64[n_sets][n_pwms] (global),
                 shards=[per set: (r0, r1)])."""
    n_pwms = len(widths)
    local_counts = np.zeros((len(sets), n_pwms), dtype=np.int64)
    all_hits, shards = [], []
    for s, (bases, offsets) in enumerate(sets):
        r0, r1 = shard_bounds(offsets, world_size)[rank]
        sb, so = take_shard(bases, offsets, r0, r1)
        hits, counts = scan_fn(pwm_values, widths, cutoffs, sb, so, strand)
        hits = dict(hits)
        hits["seq_idx"] = hits["seq_idx"] + r0
        local_counts[s] = counts
        all_hits.append(hits)
        shards.append((r0, r1))
    counts = allreduce_counts(local_counts.ravel(), device).reshape(len(sets), n_pwms)
    return {"hits": all_hits, "counts": counts, "shards": shards}


def span_shard(spans, rank, world_size):
    """A rank's share of the spans of a host-streamed genome sweep (_lib.sweep_spans): contiguous, balanced by bases.
    Windows never straddle two spans, so the per-motif window counts of the ranks simply add up (the one all-reduce) and
    a span's first_window makes every rank's window indices global."""
    sizes = np.array([sp[2] - sp[1] for sp in spans], dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(sizes)])
    a, b = shard_bounds(cum, world_size)[rank]
    return spans[a:b]


def sweep_shard(begin, end, window, stride, rank, world_size):
    """Rank's share of the windows [begin + k*stride, begin + k*stride + window), k = 0..n-1, of a sweep over [begin, end):
    (k0, k1, span_begin, span_end) -- contiguous window indices, and the base span that holds exactly those windows (spans
    of neighbouring ranks overlap by window - stride bases: each rank re-reads that much, nothing is exchanged).
    Scanning the span with ms_scan_sweep and adding k0 to seq_idx gives the rank's slice of the single-GPU result."""
    n = (end - begin - window) // stride + 1 if end - begin >= window else 0
    k0, k1 = n * rank // world_size, n * (rank + 1) // world_size
    if k1 <= k0:
        return k0, k0, begin, begin
    return k0, k1, begin + k0 * stride, begin + (k1 - 1) * stride + window


def enrichment(n_input_with_site, n_control_with_site, n_input, n_control):
    """What stats.py:18-45 computes from the reduced counts: fold change and the two one-sided
    Fisher exact p-values per motif (Bonferroni-corrected by the number of motifs)."""
    from scipy.stats import fisher_exact
    rows = []
    n_motifs = len(n_input_with_site)
    for a, c in zip(n_input_with_site.tolist(), n_control_with_site.tolist()):
        table = [[a, n_input - a], [c, n_control - c]]
        fold = a * n_control / c / n_input if c > 0 and n_input > 0 else float("nan")   # stats.py:32-35 order
        p_enrich = fisher_exact(table, alternative="greater")[1]
        p_deplete = fisher_exact(table, alternative="less")[1]
        rows.append((a, c, fold, p_enrich, p_deplete, min(min(p_enrich, p_deplete) * n_motifs, 1)))   # stats.py:40
    return rows
