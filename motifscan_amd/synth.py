"""
motifscan_amd.synth -- seeded synthetic workloads: there is no network, so no
real JASPAR file and no real genome exist on either box.

  * the motif set is package data (motifscan_amd/data/synth_jaspar579.npz): 579 JASPAR-width
    PFMs pushed through the REFERENCE's own build pipeline in the build container
    (to_ppm().to_pwm(bg), cutoffs from 10^6 background k-mers scored by the reference's c_score,
    get_score_cutoffs, around(,8)) by tests/golden/make_golden.py;
  * sequences are drawn here: iid bases from the same background, seed 1 = input set,
    seed 2 = control set, 1 % of the regions get a run of 1-50 N, 30 % of the bases are
    lower-case (soft-masked) -- exercising the reference's case folding and its
    "non-ACGT adds nothing" rule (cscore.c:92-111).
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
MOTIF_SET = os.path.join(_DATA, "synth_jaspar579.npz")

BG = np.array([0.295, 0.205, 0.205, 0.295])

WORKLOADS = {
    # name: (regions per set, region length, n_pwms, sets)   sets = 1: input only, 2: input + control
    "c2": (10_000, 500, 50, 1),          # BASELINE.json configs[1]
    "c3": (100_000, 1000, 579, 1),       # configs[2]
    "c4shard": (125_000, 500, 579, 2),   # one GPU's share of configs[3] (1M input + 1M control over 8 GPUs)
    "tiny": (512, 300, 24, 2),           # smoke / CPU-side tests
}

# configs[4] (whole-genome sweep, 3 Gbp as 200 bp windows stride 50 over 8 GPUs): one GPU's share
C5_SHARD = {"genome_bp": 375_000_000, "window": 200, "stride": 50, "n_pwms": 579}


MOTIF_SETS = {"benchmark": "synth_jaspar579.npz",          # Dirichlet(0.3) columns: every column informative (BASELINE's numbers are quoted on it)
              "lowinfo": "synth_jaspar579_lowinfo.npz"}     # JASPAR-like information profile: informative core, weak flanks, 10 % weak motifs (side workload)


def load_motif_set(n_pwms=579, p_value="1e-4", which="benchmark"):
    """(pwm_values, widths, cutoffs) of the first n_pwms synthetic JASPAR-like motifs (both sets: built through the reference's own
    to_ppm().to_pwm() and cutoff pick, tests/golden/make_golden.py)."""
    d = np.load(MOTIF_SET if which == "benchmark" else os.path.join(os.path.dirname(MOTIF_SET), MOTIF_SETS[which]))
    widths = d["widths"][:n_pwms].astype(np.int32)
    n_vals = 4 * int(widths.sum())
    keys = [str(k) for k in d["cutoff_keys"]]
    cutoffs = d["cutoffs"][:n_pwms, keys.index(p_value)].astype(np.float64)
    return d["pwm_values"][:n_vals].astype(np.float64), widths, cutoffs


def random_motif(width, seed, p_value=1e-4, n_kmers=200_000):
    """(matrix 4 x width, cutoff): one more synthetic motif made the way the fixture's were (Dirichlet(0.3) columns x depth ->
    counts -> PPM with the reference's pseudo-count -> log-odds against BG, 5 decimals; matrix.py:125-171), its cutoff the
    reference's rank pick (motif/__init__.py:393-399) over n_kmers background k-mers scored on both strands by plain numpy --
    for timing motifs wider than the fixture holds (bench workloads), not a parity vector."""
    from . import matrix
    rng = np.random.default_rng(seed)
    depth = int(rng.integers(20, 3001))
    counts = np.rint(rng.dirichlet(0.3 * np.ones(4), size=width).T * depth).astype(np.int64)
    counts[:, counts.sum(axis=0) == 0] = 1
    bg = dict(zip("ACGT", BG))
    m = matrix.PositionFrequencyMatrix(counts).to_ppm().to_pwm(bg).matrix
    codes = rng.choice(4, size=(n_kmers, width), p=BG)
    cols = np.arange(width)
    fwd = m[codes, cols].sum(axis=1)
    rev = m[3 - codes, width - 1 - cols].sum(axis=1)
    max_raw = float(np.maximum(m.max(axis=0), 0.0).sum())
    sc = np.sort(np.maximum(fwd, rev) / max_raw)[::-1]
    return m, float(np.around(sc[max(int(n_kmers * p_value) - 1, 0)], 8))


def matrices_of(pwm_values, widths):
    out, o = [], 0
    for w in widths:
        out.append(pwm_values[o:o + 4 * w].reshape(4, w))
        o += 4 * w
    return out


def make_regions(n_regions, length, seed, frac_n=0.01, frac_lower=0.30, ragged=False):
    """ASCII bases (uint8 array) + int64 offsets of n_regions synthetic regions."""
    rng = np.random.default_rng(seed)
    if ragged:
        lens = rng.integers(max(1, length // 2), length + 1, size=n_regions)
    else:
        lens = np.full(n_regions, length, dtype=np.int64)
    offsets = np.zeros(n_regions + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(lens)
    total = int(offsets[-1])
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    # inverse-CDF draw in chunks (rng.choice with p is slow and memory hungry at 10^8)
    cdf = np.cumsum(BG)
    bases = np.empty(total, dtype=np.uint8)
    step = 1 << 24
    for lo in range(0, total, step):
        hi = min(total, lo + step)
        u = rng.random(hi - lo, dtype=np.float32)
        bases[lo:hi] = letters[np.searchsorted(cdf, u, side="right").clip(0, 3)]
        lower = rng.random(hi - lo, dtype=np.float32) < frac_lower
        bases[lo:hi][lower] += 32
    n_with_n = int(round(frac_n * n_regions))
    if n_with_n:
        which = rng.choice(n_regions, size=n_with_n, replace=False)
        for r in which:
            L = int(lens[r])
            if L == 0:
                continue
            run = int(rng.integers(1, 51))
            st = int(rng.integers(0, L))
            bases[offsets[r] + st: offsets[r] + min(L, st + run)] = ord("N")
    return bases, offsets


def sweep_windows(genome_len, window, stride):
    """(chrom index, start, end) of the windows of a single-chromosome sweep."""
    starts = np.arange(0, genome_len - window + 1, stride, dtype=np.int64)
    return np.zeros(len(starts), dtype=np.int32), starts, starts + window


# ---- BASELINE configs[3] in full: 1M input + 1M control regions x 500 bp, as 8 blocks of 125k regions per set -----------
# Block b of set s is make_regions(125_000, 500, seed = 1000 * b + s + 1): exactly the shard rank b scanned in round 1
# ("c4shard" with rank = b), so the full config is their concatenation and any rank count divides it.
C4 = {"regions_per_set": 1_000_000, "length": 500, "n_pwms": 579, "n_sets": 2, "block_regions": 125_000}


def _c4_block(args):
    s, b = args
    return make_regions(C4["block_regions"], C4["length"], seed=1000 * b + s + 1)[0]


def _pool_map(fn, jobs, workers):
    """Generate blocks in worker processes (plain numpy work; started before anything touches the GPU)."""
    if os.environ.get("MS_SYNTH_WORKERS"):                  # e.g. 1 under a profiler that follows forked children
        workers = int(os.environ["MS_SYNTH_WORKERS"])
    workers = max(1, min(workers, len(jobs)))
    if workers == 1:
        return [fn(j) for j in jobs]
    import multiprocessing as mp
    pool = mp.get_context("fork").Pool(workers)
    try:
        out = pool.map(fn, jobs, chunksize=1)
        pool.close()                                        # workers leave through their normal exit path, not SIGTERM
        pool.join()
        return out
    except BaseException:
        pool.terminate()
        raise


def default_workers(world):
    """Generator processes ONE rank may start so that `world` ranks together stay within the box's cores (at most 16 each)."""
    return min(16, max(1, (os.cpu_count() or 1) // max(int(world), 1)))


def c4_shard(rank=0, world=1, workers=None, regions_per_set=None):
    """Rank's contiguous share of the full configs[3] workload: dict like workload() with sets = [(bases, offsets)] of the
    rank's regions [r0, r1) of each set, plus "shard" = (r0, r1).  Only the blocks the share overlaps are generated."""
    R = int(regions_per_set or C4["regions_per_set"])
    L, B = C4["length"], C4["block_regions"]
    r0, r1 = R * rank // world, R * (rank + 1) // world
    blocks = list(range(r0 // B, (r1 + B - 1) // B)) if r1 > r0 else []
    jobs = [(s, b) for s in range(C4["n_sets"]) for b in blocks]
    if workers is None:
        workers = default_workers(world)
    made = dict(zip(jobs, _pool_map(_c4_block, jobs, workers)))
    vals, widths, cutoffs = load_motif_set(C4["n_pwms"])
    sets = []
    for s in range(C4["n_sets"]):
        parts = []
        for b in blocks:
            lo, hi = max(r0, b * B) - b * B, min(r1, (b + 1) * B) - b * B
            parts.append(made[(s, b)][lo * L:hi * L])
        bases = np.concatenate(parts) if len(parts) > 1 else (parts[0] if parts else np.zeros(0, np.uint8))
        sets.append((bases, np.arange(r1 - r0 + 1, dtype=np.int64) * L))
    return {"name": "c4", "pwm_values": vals, "widths": widths, "cutoffs": cutoffs, "sets": sets, "shard": (r0, r1),
            "units": sum(int(o[-1]) for _, o in sets) * C4["n_pwms"], "units_total": C4["n_sets"] * R * L * C4["n_pwms"],
            "n_regions": r1 - r0, "n_regions_total": R, "length": L, "n_pwms": C4["n_pwms"]}


# ---- BASELINE configs[4] in full: a multi-chromosome genome on the host, swept as 200 bp windows stride 50 ---------------
C5 = {"genome_bp": 3_000_000_000, "n_chroms": 24, "window": 200, "stride": 50, "n_pwms": 579, "block_bp": 62_500_000}


def c5_chrom_lengths(genome_bp=None, n_chroms=None):
    """Chromosome lengths of the synthetic genome: human-like spread (largest ~5x the smallest), summing to genome_bp."""
    G = int(genome_bp or C5["genome_bp"])
    n = int(n_chroms or C5["n_chroms"])
    w = np.linspace(5.0, 1.0, n)
    lens = np.floor(w / w.sum() * G).astype(np.int64)
    lens[0] += G - int(lens.sum())
    return lens


def _c5_block(args):
    ch, k, n = args
    b = make_regions(1, n, seed=7000 + 131 * ch + k, frac_n=0.0)[0]
    rng = np.random.default_rng(9000 + 131 * ch + k)
    for _ in range(max(1, n // 2_000_000)):                         # assembly gaps: runs of N
        st = int(rng.integers(0, max(1, n - 5000)))
        b[st:st + int(rng.integers(50, 5000))] = ord("N")
    return b


def c5_genome(chroms, lens, workers=None, out=None):
    """The listed chromosomes (indices into lens) as uint8 arrays, generated block-wise in worker processes.  `out`:
    optional dict chrom -> preallocated uint8 array (e.g. pinned host memory) to fill instead of allocating."""
    jobs = []
    for ch in chroms:
        L, B = int(lens[ch]), C5["block_bp"]
        jobs += [(ch, k, min(B, L - k * B)) for k in range((L + B - 1) // B)]
    if workers is None:
        workers = min(16, os.cpu_count() or 1)
    made = _pool_map(_c5_block, jobs, workers)
    res = {}
    for ch in chroms:
        parts = [m for (c, k, n), m in zip(jobs, made) if c == ch]
        if out is not None:
            o = 0
            for m in parts:
                out[ch][o:o + len(m)] = m
                o += len(m)
            res[ch] = out[ch]
        else:
            res[ch] = np.concatenate(parts) if len(parts) > 1 else (parts[0] if parts else np.zeros(0, np.uint8))
    return res


def _c5_block_to_file(args):
    path, ch, k, n, off = args
    mm = np.memmap(path, dtype=np.uint8, mode="r+")
    mm[off:off + n] = _c5_block((ch, k, n))
    mm.flush()
    return n


def c5_genome_to_dir(dirpath, lens, chroms=None, workers=None):
    """The same chromosomes as c5_genome() (block for block, seed for seed), written straight into one flat uint8 file per
    chromosome under dirpath (chr<i>.u8): the workers fill disjoint slices of a shared mapping, nothing is pickled back.  Meant to
    run in a FRESH python process (a test that has already initialised the GPU starts it as a child, never forks itself).
    Returns the file paths by chromosome index."""
    chroms = list(range(len(lens))) if chroms is None else list(chroms)
    jobs, paths = [], {}
    for ch in chroms:
        L, B = int(lens[ch]), C5["block_bp"]
        path = os.path.join(dirpath, f"chr{ch}.u8")
        np.memmap(path, dtype=np.uint8, mode="w+", shape=(max(L, 1),)).flush()
        paths[ch] = path
        jobs += [(path, ch, k, min(B, L - k * B), k * B) for k in range((L + B - 1) // B)]
    if workers is None:
        workers = min(32, os.cpu_count() or 1)
    _pool_map(_c5_block_to_file, jobs, workers)
    return paths


def workload(name, rank=0):
    """Returns dict(pwm_values, widths, cutoffs, sets=[(bases, offsets), ...], units) for a named
    workload; rank shifts the sequence seeds so every GPU scans different regions."""
    if name == "c5shard":
        c = C5_SHARD
        vals, widths, cutoffs = load_motif_set(c["n_pwms"])
        genome, _ = make_regions(1, c["genome_bp"], seed=5000 + rank, frac_n=0.0)
        rng = np.random.default_rng(6000 + rank)
        for _ in range(c["genome_bp"] // 2_000_000):               # assembly gaps: runs of N
            st = int(rng.integers(0, c["genome_bp"] - 5000))
            genome[st:st + int(rng.integers(50, 5000))] = ord("N")
        win = sweep_windows(c["genome_bp"], c["window"], c["stride"])
        n_win = len(win[0])
        return {"name": name, "pwm_values": vals, "widths": widths, "cutoffs": cutoffs, "sets": [], "genome": genome,
                "windows": win, "units": n_win * c["window"] * c["n_pwms"], "n_regions": n_win, "length": c["window"],
                "n_pwms": c["n_pwms"]}
    n_regions, length, n_pwms, n_sets = WORKLOADS[name]
    vals, widths, cutoffs = load_motif_set(n_pwms)
    sets = [make_regions(n_regions, length, seed=1000 * rank + s + 1) for s in range(n_sets)]
    units = sum(int(off[-1]) for _, off in sets) * n_pwms          # metric unit: region bp x motifs
    return {"name": name, "pwm_values": vals, "widths": widths, "cutoffs": cutoffs, "sets": sets,
            "units": units, "n_regions": n_regions, "length": length, "n_pwms": n_pwms}
