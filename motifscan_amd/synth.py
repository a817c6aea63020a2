"""
motifscan_amd.synth -- seeded synthetic workloads: there is no network, so no
real JASPAR file and no real genome exist on either box.

  * the motif set is a committed fixture (tests/golden/synth_jaspar579.npz): 579 JASPAR-width
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

_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

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


def load_motif_set(n_pwms=579, p_value="1e-4"):
    """(pwm_values, widths, cutoffs) of the first n_pwms synthetic JASPAR-like motifs."""
    d = np.load(os.path.join(_GOLDEN, "synth_jaspar579.npz"))
    widths = d["widths"][:n_pwms].astype(np.int32)
    n_vals = 4 * int(widths.sum())
    keys = [str(k) for k in d["cutoff_keys"]]
    cutoffs = d["cutoffs"][:n_pwms, keys.index(p_value)].astype(np.float64)
    return d["pwm_values"][:n_vals].astype(np.float64), widths, cutoffs


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
