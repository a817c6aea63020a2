"""
Randomised parity fuzzer: HIP path (through the C-ABI) vs the oracle on seeded random cases built
to sit ON the decision boundary of cscore.c:360-389 (`score / max_raw - cutoff >= -1e-10`):

  * matrices with few distinct values (integers, halves, one repeated value) so many windows tie;
  * cutoffs placed exactly on attainable scores (k / max_raw), one ulp either side of them, and
    1e-10 either side (the reference's own slack);
  * widths 1..66 (every k-block class of the pre-filter, and the all-fp64 kernel past 63 columns), all-negative matrices (max_raw == 0),
    huge / tiny magnitudes, cutoffs <= 0 (everything hits) and > 1 (nothing can);
  * sequences with N runs, lower case, other IUPAC letters, empty and shorter-than-W regions;
  * (round 6) every case also through MS_SCAN_COUNTS_ONLY and through a two-batch stream with the 12-byte copy-out.

It lives under tests/ because it uses the oracle (test infrastructure).  Run on the GPU box:
    python tests/fuzz_parity.py --cases 200 --seed 0
`tests/test_gpu_parity.py::test_fuzz_decision_boundary` runs a few cases of it in the GPU suite.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def random_matrix(rng, w):
    kind = rng.integers(0, 8)
    if kind == 0:                                   # small integers: masses of exact ties
        m = rng.integers(-3, 4, size=(4, w)).astype(np.float64)
    elif kind == 1:                                 # halves / quarters (exact in binary)
        m = rng.integers(-8, 9, size=(4, w)) / 4.0
    elif kind == 2:                                 # log-odds-like, rounded as the reference's files are
        p = rng.dirichlet(np.full(4, rng.choice([0.2, 1.0, 5.0])), size=w).T
        m = np.round(np.log2(np.maximum(p, 1e-4) / 0.25), 6)
    elif kind == 3:                                 # huge magnitudes
        m = rng.normal(0, 1, size=(4, w)) * 10.0 ** rng.integers(2, 7)
    elif kind == 4:                                 # tiny magnitudes
        m = rng.normal(0, 1, size=(4, w)) * 10.0 ** -rng.integers(3, 9)
    elif kind == 5:                                 # all negative: max_raw == 0
        m = -np.abs(rng.normal(0, 1, size=(4, w))) - 0.01
    elif kind == 6:                                 # one informative column, the rest flat
        m = np.zeros((4, w))
        m[:, rng.integers(0, w)] = rng.normal(0, 2, size=4)
    else:                                           # one strong base per column, mixed penalties
        m = np.full((4, w), -float(rng.integers(1, 6)))
        m[rng.integers(0, 4, size=w), np.arange(w)] = float(rng.integers(1, 3))
    return np.ascontiguousarray(m, dtype=np.float64)


def max_raw_of(m):
    return float(np.sum(np.maximum(m.max(axis=0), 0.0)))


def random_sequences(rng, n, max_len):
    out = []
    for _ in range(n):
        kind = rng.integers(0, 10)
        L = int(rng.integers(0, max_len + 1))
        if kind == 0:
            s = ""
        elif kind == 1:
            s = "ACGT"[rng.integers(0, 4)] * L
        elif kind == 2:
            unit = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 5))))
            s = (unit * (L // len(unit) + 1))[:L]
        else:
            p = np.array([.24, .24, .24, .24, .03, .01])
            s = "".join(rng.choice(list("ACGTNR"), p=p, size=L))
            if kind == 3 and L > 10:
                a = int(rng.integers(0, L - 5))
                s = s[:a] + "N" * int(rng.integers(1, 40)) + s[a:]
            if kind == 4:
                s = s.lower()
            elif kind == 5:
                s = "".join(c.lower() if rng.random() < 0.3 else c for c in s)
        out.append(s)
    return out


def attainable_cutoff(rng, m, seqs):
    """A cutoff sitting on (or a hair beside) the ratio of a window that really occurs."""
    w = m.shape[1]
    mr = max_raw_of(m)
    cands = [s for s in seqs if len(s) >= w and set(s.upper()) <= set("ACGT")]
    if mr <= 0 or not cands or rng.random() < 0.15:
        return float(rng.choice([-0.2, 0.0, 0.3, 0.8, 1.0, 1.0 + 1e-12, 1.3]))
    s = cands[rng.integers(0, len(cands))].upper()
    a = int(rng.integers(0, len(s) - w + 1))
    score = 0.0
    for c in range(w):                              # same summation order as cscore.c:352-358
        score += m["ACGT".index(s[a + c]), c]
    ratio = score / mr
    nudge = rng.integers(0, 7)
    if nudge == 0:
        return ratio
    if nudge == 1:
        return float(np.nextafter(ratio, np.inf))
    if nudge == 2:
        return float(np.nextafter(ratio, -np.inf))
    if nudge == 3:
        return ratio + 1e-10
    if nudge == 4:
        return ratio + 1.0000001e-10
    if nudge == 5:
        return ratio + 0.9999999e-10
    return float(np.round(ratio, 8))                # what `motif --build` writes (np.around(, 8))


def make_case(seed):
    rng = np.random.default_rng(seed)
    n_motifs = int(rng.choice([1, 2, 5, 13, 40, 97, 200]))
    wmax = int(rng.choice([8, 16, 32, 40, 66]))            # 66: row tiles of 3 and 4 k-blocks, and the all-fp64 kernel past 63 columns
    mats = [random_matrix(rng, int(rng.integers(1, wmax + 1))) for _ in range(n_motifs)]
    seqs = random_sequences(rng, int(rng.choice([1, 7, 60, 300])), int(rng.choice([20, 150, 700])))
    cutoffs = np.array([attainable_cutoff(rng, m, seqs) for m in mats], dtype=np.float64)
    strand = int(rng.integers(1, 4))
    return mats, cutoffs, seqs, strand


def check_dedup_and_score(seed, oracle, _lib, mats, widths, seqs, strand, want, pw, res):
    """Device de-dup (ms_result_dedup) vs the oracle's restatement of scanner.py:156-193 on the nested
    lists (cases small enough for Python lists), and ms_score vs oracle_score on the regions long
    enough for every motif (shorter ones read out of bounds in the reference, cscore.c:196-204)."""
    if len(want["pos"]) <= 150_000:
        off = want["motif_offsets"]
        cols = [want[k].tolist() for k in ("seq_idx", "pos", "score", "strand")]
        nested = [[[cols[0][k], cols[1][k], cols[2][k], cols[3][k]] for k in range(off[p], off[p + 1])]
                  for p in range(len(widths))]
        dd = oracle.deduplicate_motif_sites(oracle.make_motif_sites(nested, [0] * len(seqs)), widths.tolist())
        flat = [(p, r, s.start, s.score, 1 if s.strand == "+" else 2)
                for p, per in enumerate(dd) for r, sites in enumerate(per) for s in sites]
        res.dedup(pw)
        d = res.hits()
        motif = np.repeat(np.arange(len(widths)), np.diff(d["motif_offsets"]))
        mine = list(zip(motif.tolist(), d["seq_idx"].tolist(), d["pos"].tolist(), d["score"].tolist(),
                        d["strand"].astype(np.int32).tolist()))
        if mine != flat:
            return f"seed {seed}: de-duplicated sites differ ({len(mine)} vs {len(flat)})"
    wmax = int(widths.max())
    long_seqs = [s for s in seqs if len(s) >= wmax]
    if long_seqs:
        raw = "".join(long_seqs).encode()
        offsets = np.concatenate([[0], np.cumsum([len(s) for s in long_seqs])]).astype(np.int64)
        vals = np.concatenate([m.ravel() for m in mats])
        sq = _lib.SeqSet(raw, offsets)
        try:
            got = _lib.score(pw, sq, strand)
        finally:
            sq.close()
        ref = oracle.score_arrays(vals, widths, raw, offsets, strand, 4)
        if not np.array_equal(got, ref, equal_nan=True):
            return f"seed {seed}: c_score differs"
    return None


def check_round6_paths(seed, _lib, raw, offsets, strand, want, pw, sq):
    """Round 6's paths on the same case: MS_SCAN_COUNTS_ONLY (n_hits, per-motif site numbers, per-motif region counts from the UNORDERED hits,
    or from the ordered path where the flag map does not apply) and the batch stream with the 12-byte copy-out (two batches cut at a random
    region; packing on the scan stage; the second batch counts-only every other seed)."""
    n_motifs = len(want["motif_offsets"]) - 1
    pair = np.unique((np.repeat(np.arange(n_motifs), np.diff(want["motif_offsets"])).astype(np.int64) << 32) | want["seq_idx"])
    want_regions = np.bincount(pair >> 32, minlength=n_motifs)
    co = _lib.scan(pw, sq, strand, _lib.MS_SCAN_COUNTS_ONLY)
    try:
        if co.n_hits != len(want["pos"]) or not np.array_equal(co.motif_offsets, want["motif_offsets"]) or not np.array_equal(co.region_counts(), want_regions):
            return f"seed {seed}: counts-only scan differs"
    finally:
        co.close()
    R = len(offsets) - 1
    if R < 2:
        return None
    cut = 1 + seed % (R - 1)
    second_counts_only = seed % 2 == 1
    b = np.frombuffer(raw, dtype=np.uint8)
    batches = [(b[:int(offsets[cut])], offsets[:cut + 1].copy(), False), (b[int(offsets[cut]):], offsets[cut:] - offsets[cut], second_counts_only)]
    parts, counts = [], np.zeros(n_motifs, dtype=np.int64)
    for (bb, oo, co_), start, res in zip(batches, (0, cut), _lib.scan_stream(pw, iter(batches), strand, packed=12)):
        counts += res.region_counts()
        if not co_:
            parts.append((res.hits(packed=True), start))
        res.close()
    if not np.array_equal(counts, want_regions):
        return f"seed {seed}: the stream's region counts differ"
    merged = _lib.merge_hits(parts, n_motifs)
    sel = np.ones(len(want["pos"]), dtype=bool) if not second_counts_only else want["seq_idx"] < cut
    for k in ("seq_idx", "pos", "score"):
        if not np.array_equal(merged[k], want[k][sel]):
            return f"seed {seed}: the 12-byte stream's {k} differ"
    if not np.array_equal(merged["strand"].astype(np.int32), want["strand"].astype(np.int32)[sel]):
        return f"seed {seed}: the 12-byte stream's strands differ"
    return None


def run_case(seed, oracle, _lib):
    mats, cutoffs, seqs, strand = make_case(seed)
    vals = np.concatenate([m.ravel() for m in mats])
    widths = np.array([m.shape[1] for m in mats], dtype=np.int32)
    raw = "".join(seqs).encode()
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
    want = oracle.scan_arrays(vals, widths, cutoffs, raw, offsets, strand, 4)
    pw, sq = _lib.PwmSet(vals, widths, cutoffs), _lib.SeqSet(raw, offsets)
    res = _lib.scan(pw, sq, strand)
    try:
        got = {k: v.copy() for k, v in res.hits().items()}
        st = res.stats()
        extra = check_round6_paths(seed, _lib, raw, offsets, strand, want, pw, sq)
        extra = extra or check_dedup_and_score(seed, oracle, _lib, mats, widths, seqs, strand, want, pw, res)
    finally:
        res.close()
        sq.close()
        pw.close()
    if extra:
        return False, extra, st
    for k in ("motif_offsets", "seq_idx", "pos"):
        if not np.array_equal(got[k], want[k]):
            return False, f"seed {seed}: {k} differs ({len(got['pos'])} vs {len(want['pos'])} hits)", st
    if not np.array_equal(got["strand"].astype(np.int32), want["strand"].astype(np.int32)):
        return False, f"seed {seed}: strand differs", st
    if not np.array_equal(got["score"], want["score"]):
        return False, f"seed {seed}: scores differ (max abs {np.max(np.abs(got['score'] - want['score']))})", st
    return True, len(want["pos"]), st


def run_sweep_case(seed, oracle, _lib):
    """ms_scan_sweep (every base scored once, hits handed to the windows that hold them) vs the oracle over the same
    windows as separate regions; random window / stride incl. stride > window and windows narrower than motifs."""
    rng = np.random.default_rng(1_000_003 * 7 + seed)
    n_motifs = int(rng.choice([1, 3, 20, 60]))
    mats = [random_matrix(rng, int(rng.integers(1, 34))) for _ in range(n_motifs)]
    L = int(rng.choice([50, 400, 3000]))
    kind = rng.integers(0, 4)
    if kind == 0:
        unit = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 4))))
        chrom = (unit * (L // len(unit) + 1))[:L]                       # low complexity: dense neighbourhoods
    else:
        chrom = "".join(rng.choice(list("ACGTNacgt"), p=[.22, .22, .22, .22, .04, .02, .02, .02, .02], size=L))
    window = int(rng.choice([5, 12, 30, 64, 200]))
    stride = int(rng.choice([1, 3, 7, 25, 50, 300]))
    begin = int(rng.integers(0, max(1, L // 4)))
    end = int(rng.integers(begin, L + 1))
    n_win = (end - begin - window) // stride + 1 if end - begin >= window else 0
    if n_win > 20000:
        stride = max(stride, (end - begin) // 20000 + 1)
        n_win = (end - begin - window) // stride + 1
    seqs = [chrom[begin + k * stride: begin + k * stride + window] for k in range(n_win)]
    cutoffs = np.array([attainable_cutoff(rng, m, seqs[:50] + [chrom]) for m in mats], dtype=np.float64)
    strand = int(rng.integers(1, 4))
    vals = np.concatenate([m.ravel() for m in mats])
    widths = np.array([m.shape[1] for m in mats], dtype=np.int32)
    raw = "".join(seqs).encode()
    offsets = np.arange(n_win + 1, dtype=np.int64) * window
    want = oracle.scan_arrays(vals, widths, cutoffs, raw, offsets, strand, 4)
    genome = _lib.ResidentGenome({"x": "ACGT" * 3, "chr": chrom})
    pw = _lib.PwmSet(vals, widths, cutoffs)
    res = _lib.scan_sweep(pw, genome, "chr", begin, end, window, stride, strand)
    try:
        got = {k: v.copy() for k, v in res.hits().items()}
        counts = res.region_counts()
    finally:
        res.close()
        pw.close()
        genome.close()
    for k in ("motif_offsets", "seq_idx", "pos", "score"):
        if not np.array_equal(got[k], want[k]):
            return False, f"sweep seed {seed}: {k} differs ({len(got['pos'])} vs {len(want['pos'])} sites)"
    if not np.array_equal(got["strand"].astype(np.int32), want["strand"].astype(np.int32)):
        return False, f"sweep seed {seed}: strand differs"
    pair = np.unique((np.repeat(np.arange(n_motifs), np.diff(want["motif_offsets"])).astype(np.int64) << 32) | want["seq_idx"])
    if not np.array_equal(counts, np.bincount(pair >> 32, minlength=n_motifs)):
        return False, f"sweep seed {seed}: window counts differ"
    return True, len(want["pos"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--sweep", action="store_true", help="fuzz ms_scan_sweep instead of ms_scan")
    a = ap.parse_args()
    from oracle import oracle
    oracle.build()
    from motifscan_amd import _lib
    _lib.set_device(0)
    if a.sweep:
        bad, total = 0, 0
        for k in range(a.cases):
            ok, info = run_sweep_case(a.seed + k, oracle, _lib)
            if not ok:
                bad += 1
                print("MISMATCH", info, flush=True)
            else:
                total += info
        print(f"sweep fuzz: {a.cases} cases from seed {a.seed}: {bad} mismatches, {total} sites compared")
        return 1 if bad else 0
    bad, total_hits, fast, exact = 0, 0, 0, 0
    for k in range(a.cases):
        ok, info, st = run_case(a.seed + k, oracle, _lib)
        if not ok:
            bad += 1
            print("MISMATCH", info, flush=True)
        else:
            total_hits += info
        fast += st["n_pwms"] - st["n_pwms_exact"]
        exact += st["n_pwms_exact"]
    print(f"fuzz: {a.cases} cases from seed {a.seed}: {bad} mismatches, {total_hits} hits compared, "
          f"{fast} motifs through the pre-filter, {exact} through the exact-only kernel")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
