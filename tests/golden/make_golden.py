#!/usr/bin/env python3
"""
tests/golden/make_golden.py -- generate the committed golden vectors by running the REAL
reference (MotifScan 1.3.0) in the build container.  Nothing here runs on the GPU box and
nothing here is product code.

How the reference is reached (no reference source enters this repo):
  * /root/reference is put on sys.path and imported as the `motifscan` package;
  * its one native module, motifscan.motif.cscore, is the unmodified cscore.c compiled by
    oracle/Makefile into oracle/_ref/ and registered under its usual module name;
  * `pysam` (htslib binding, absent from this image) is imported by motifscan.genome at
    module load only to open FASTA files.  An EMPTY placeholder module satisfies that
    import; no pysam function is ever called: Scanner only needs an object with
    `chrom_sizes` and `fetch_sequence(chrom, start, end)` (scanner.py:71-87), which is
    provided here by a dict-backed class over the fixture's chromosome strings
    (0-based half-open slices, which is what the reference's own test pins:
    tests/test_scanner.py:17,22 -> 'TtC' / 'aTtC').

Outputs (all under tests/golden/):
  ref_small.json        G1/G2/G6/G7: known answers of the reference's tests re-run, toy-genome
                        Scanner results, edge cases, matrix numerics, de-dup cases
  ref_genome.json       N3 (round 6): the toy genome's FASTA / .fai as data, the reference tests' fetch literals, every slice
  ref_random.npz        G3/G4/G5: seeded random PWMs built by the reference's PFM->PPM->PWM,
                        cutoffs from c_score + get_score_cutoffs, sequences with N / soft-mask /
                        IUPAC / short and empty entries; full hit lists for strands 1,2,3;
                        Scanner.scan_motifs with and without de-dup; c_score on k-mers
  ../../motifscan_amd/data/synth_jaspar579.npz   the benchmark motif set (package data): 579 JASPAR-width synthetic PFMs pushed through the
                        reference's build pipeline (to_ppm().to_pwm(bg), cutoffs from 10^6 background
                        k-mers scored by the reference's c_score, get_score_cutoffs, around(,8))

Usage:  python3 tests/golden/make_golden.py [--skip-579]
"""
import argparse
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

sys.path.insert(0, ROOT)
from oracle import oracle as _oracle  # noqa: E402  (only to locate/load oracle/_ref)


def import_reference():
    ext = _oracle.load_reference_ext()
    if ext is None:
        raise SystemExit("oracle/_ref is not built: run `make -C oracle` in the build container")
    sys.modules.setdefault("pysam", types.ModuleType("pysam"))   # empty placeholder, see docstring
    sys.path.insert(0, REF)
    import motifscan  # noqa: F401
    import motifscan.motif as mm
    sys.modules["motifscan.motif.cscore"] = ext
    mm.cscore = ext
    import motifscan.scanner as sc
    from motifscan.motif.matrix import (PositionFrequencyMatrix, PositionProbabilityMatrix,
                                        PositionWeightMatrix)
    from motifscan.motif import get_score_cutoffs
    from motifscan.region import GenomicRegion
    return dict(ext=ext, scanner=sc, PFM=PositionFrequencyMatrix, PPM=PositionProbabilityMatrix,
                PWM=PositionWeightMatrix, get_score_cutoffs=get_score_cutoffs, GenomicRegion=GenomicRegion,
                version=motifscan.__version__)


class DictGenome:
    """Object with the two members Scanner uses (scanner.py:81-87)."""

    def __init__(self, chroms):
        self._c = dict(chroms)
        self.chrom_sizes = {k: len(v) for k, v in self._c.items()}

    def fetch_sequence(self, chrom, start, end):
        return self._c[chrom][start:end]


def read_fasta(path):
    chroms, name = {}, None
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if line.startswith(">"):
                name = line[1:].split()[0]
                chroms[name] = ""
            elif name is not None:
                chroms[name] += line
    return chroms


def sites_to_rows(motif_sites):
    rows = []
    for p, per_pwm in enumerate(motif_sites):
        for r, sites in enumerate(per_pwm):
            for s in sites:
                rows.append([p, r, int(s.start), float(s.score), s.strand])
    return rows


# --------------------------------------------------------------------------- small --

def make_small(R):
    ext, sc = R["ext"], R["scanner"]
    out = {"reference_version": R["version"]}

    # G1: the reference's own known-answer inputs (tests/test_motif_score.py:6-32), re-run
    m3 = [[[1.35, 0.21, -5.23], [0.07, -0.21, 0.6], [2.15, 2.22, -0.84], [-2.64, -1.89, 5.47]]]
    out["G1"] = {
        "matrix": m3,
        "score_seqs": ["NNN", "AGT", "ANT", "CTA"],
        "score": {str(s): ext.c_score(m3, ["NNN", "AGT", "ANT", "CTA"], s, 1) for s in (1, 2, 3)},
        "scan_seqs": ["NNNAG", "TANTCTA"],
        "scan_cutoffs": [0.2],
        "scan": {str(s): ext.c_scan_motif(m3, [0.2], ["NNNAG", "TANTCTA"], s, 1) for s in (1, 2, 3)},
    }

    # G2: Scanner on the reference's toy genome (tests/data/genomes/test/test.fa, a data file)
    chroms = read_fasta(os.path.join(REF, "tests/data/genomes/test/test.fa"))
    genome = DictGenome(chroms)
    GR = R["GenomicRegion"]
    regions = [GR(chrom="chr1", start=2, end=5)]
    s0 = sc.Scanner(genome=genome, regions=regions, window_size=0)
    s4 = sc.Scanner(genome=genome, regions=regions, window_size=4, strand="+")
    assert s0.sequences == ["TtC"] and s4.sequences == ["aTtC"]       # tests/test_scanner.py:17,22
    pwm = R["PWM"]([[1, 0], [0, 1], [0, 0], [1, 0]], cutoffs={"1e-3": 0.5, "1e-4": 1})
    g2 = {"chroms": chroms, "region": ["chr1", 2, 5], "pwm": [[1, 0], [0, 1], [0, 0], [1, 0]],
          "cutoffs": {"1e-3": 0.5, "1e-4": 1},
          "extract": {"w0": [s0.sequences, s0.seq_starts, s0.seq_ends],
                      "w4": [s4.sequences, s4.seq_starts, s4.seq_ends]},
          "cases": []}
    for p_value, dup in (("1e-4", True), ("1e-3", True), ("1e-3", False)):
        s = sc.Scanner(genome=genome, regions=regions, window_size=4, p_value=p_value, remove_dup=dup)
        g2["cases"].append({"p_value": p_value, "remove_dup": dup, "window_size": 4,
                            "sites": sites_to_rows(s.scan_motifs([pwm]))})
    # whole-chromosome scan of the toy genome with the reference's built PWM file values
    toy = []
    with open(os.path.join(REF, "tests/data/motifs/test/test_pwms.motifscan")) as fh:
        cur = None
        for line in fh:
            line = line.strip()
            if line.startswith(">"):
                cur = {"name": line[1:].split("\t")[1], "matrix": [], "cutoffs": {}}
                toy.append(cur)
            elif line[:1] in "ACGT" and "[" in line:
                cur["matrix"].append([float(x) for x in line[line.index("[") + 1:line.index("]")].split()])
            elif line.startswith("Cutoff_p"):
                k, v = line.split("\t")
                cur["cutoffs"][k[len("Cutoff_p"):]] = float(v)
    whole = [GR(chrom=c, start=0, end=len(s)) for c, s in chroms.items()]
    g2["toy_pwms"] = toy
    g2["whole_regions"] = [[r.chrom, r.start, r.end] for r in whole]
    g2["whole"] = []
    for p_value in ("1e-2", "1e-3"):
        pw = [R["PWM"](t["matrix"], cutoffs=t["cutoffs"]) for t in toy]
        s = sc.Scanner(genome=genome, regions=whole, window_size=0, p_value=p_value)
        g2["whole"].append({"p_value": p_value, "sites": sites_to_rows(s.scan_motifs(pw))})
    out["G2"] = g2

    # G6: edge cases / quirks, each run through the real c_scan_motif / c_score
    q4 = [[[-1, 2], [-2, 1], [-3, .5], [-4, .1]]]
    pal = [[[2.0, -1.0, -1.0, -3.0], [-1.0, 2.0, -3.0, -1.0], [-1.0, -3.0, 2.0, -1.0], [-3.0, -1.0, -1.0, 2.0]]]
    edge = []

    def scan_case(name, pwms, cutoffs, seqs, strand):
        edge.append({"name": name, "kind": "scan", "pwms": pwms, "cutoffs": cutoffs, "seqs": seqs,
                     "strand": strand, "out": ext.c_scan_motif(pwms, cutoffs, seqs, strand, 1)})

    def score_case(name, pwms, seqs, strand):
        edge.append({"name": name, "kind": "score", "pwms": pwms, "seqs": seqs, "strand": strand,
                     "out": ext.c_score(pwms, seqs, strand, 1)})

    scan_case("Q1_iupac_and_case", m3, [-10.0], ["ARY", "NNN", "agt", "AgT"], 1)
    scan_case("Q2_all_N_window_hits_when_cutoff_negative", m3, [-0.05], ["NNNNN", "ANNNA"], 3)
    scan_case("Q2_all_N_window_no_hit_when_cutoff_positive", m3, [0.05], ["NNNNN"], 3)
    scan_case("Q3_short_and_empty", m3, [-10.0], ["", "A", "AG", "AGT", "AGTC"], 3)
    scan_case("Q4_all_negative_column_maxraw_clamped", q4, [-10.0], ["ACGTAC"], 3)
    score_case("Q4_score", q4, ["AC", "CA", "GT", "NN"], 3)
    base = ext.c_scan_motif(m3, [-10.0], ["AGT"], 1, 1)[0][0][2]
    scan_case("Q6_tolerance_inside", m3, [base + 5e-11], ["AGT"], 1)
    scan_case("Q6_tolerance_outside", m3, [base + 2e-10], ["AGT"], 1)
    scan_case("Q7_perfect_match_both_strands", pal, [1.0], ["ACGT", "TTACGTTT"], 3)
    for s in (1, 2, 3):
        scan_case(f"Q8_strand_{s}", m3, [0.2], ["NNNAG", "TANTCTA", "GGTACCAGT"], s)
    scan_case("two_pwms_different_width", m3 + q4, [0.2, 0.5], ["TANTCTAGGAC", "ac"], 3)
    scan_case("empty_pwm_list", [], [], ["ACGT"], 3)
    scan_case("empty_seq_list", m3, [0.2], [], 3)
    out["G6"] = edge

    # G7: matrix numerics (tests/test_motif_matrix.py:37-116), full-precision values from the reference
    PFM, PPM, PWM = R["PFM"], R["PPM"], R["PWM"]
    pfm_vals = [[1, 1], [1, 2], [1, 2], [1, 0]]
    ppm_vals = [[0.2, 0.2], [0.2, 0.2], [0.3, 0.6], [0.3, 0]]
    bg = {"A": 0.22, "C": 0.23, "G": 0.28, "T": 0.27}
    ppm_n = PPM(ppm_vals)
    ppm_n.normalize(pseudo=0.001)
    w = PWM(m3[0])
    out["G7"] = {
        "pfm": pfm_vals,
        "pfm_to_ppm_raw": PFM(pfm_vals).to_ppm(normalize=False).matrix.tolist(),
        "pfm_to_ppm_norm": PFM(pfm_vals).to_ppm(normalize=True, pseudo=0.001).matrix.tolist(),
        "ppm": ppm_vals,
        "ppm_normalized": ppm_n.matrix.tolist(),
        "ppm_to_pwm_default_bg": ppm_n.to_pwm().matrix.tolist(),
        "bg": bg,
        "ppm_to_pwm_bg": ppm_n.to_pwm(bg_freq=bg).matrix.tolist(),
        "pwm": m3[0],
        "max_raw_score": float(w.max_raw_score),
        "min_raw_score": float(w.min_raw_score),
        "score": {s: float(w.score(s)) for s in ("NNN", "AGT", "ANT", "CTA", "agt")},
    }

    # de-dup semantics (tests/test_scanner.py:57-73 plus ties / chains)
    out["dedup"] = dedup_cases(sc)

    out["N4"] = make_formats(R, genome, GR, toy, chroms)

    with open(os.path.join(HERE, "ref_small.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote ref_small.json")


def dedup_cases(sc):
    """scanner.py:156-193 run on hand-made lists -- start-sorted ones (what a scan produces) and lists in ARBITRARY order (what a
    caller may hand over: the reference walks each strand's sites in the order given, so a negative distance also counts as
    'closer than the motif length')."""
    MS = sc.MotifSite
    rng = np.random.default_rng(77)
    shuffled = []
    for k in range(6):
        n = int(rng.integers(5, 40))
        sites = [(int(rng.integers(0, 60)), float(np.round(rng.random(), 3)), "+-"[int(rng.integers(0, 2))]) for _ in range(n)]
        shuffled.append((f"unsorted_random_{k}", sites, int(rng.integers(2, 12))))
    cases = []
    for name, sites, length in (
            ("reference_test", [(1, 1, "+"), (3, 0.8, "+"), (1, 1, "-"), (2, 3, "-"), (5, 1, "+")], 3),
            ("tie_keeps_earlier", [(1, 0.5, "+"), (2, 0.5, "+"), (3, 0.5, "+"), (10, 0.5, "+")], 3),
            ("chain_rising", [(1, 0.1, "+"), (2, 0.2, "+"), (3, 0.3, "+"), (4, 0.4, "+"), (9, 0.1, "+")], 4),
            ("chain_falling", [(1, 0.4, "-"), (2, 0.3, "-"), (3, 0.2, "-"), (6, 0.9, "-")], 4),
            ("interleaved_strands", [(1, 0.4, "+"), (1, 0.9, "-"), (2, 0.5, "+"), (2, 0.1, "-"), (8, 1, "+")], 5),
            ("unsorted_descending", [(9, 0.2, "+"), (7, 0.9, "+"), (1, 0.5, "+"), (3, 0.1, "-"), (2, 0.8, "-")], 3),
            ("unsorted_far_apart_backwards", [(50, 0.3, "+"), (10, 0.2, "+"), (30, 0.9, "+"), (31, 0.1, "+")], 4),
            ("unsorted_equal_starts", [(5, 0.1, "-"), (5, 0.7, "-"), (4, 0.7, "-"), (20, 0.2, "+"), (5, 0.3, "+")], 2)) + tuple(shuffled):
        ms = [[[MS(*s) for s in sites]]]
        res = sc.deduplicate_motif_sites(ms, [length])
        cases.append({"name": name, "sites": [list(s) for s in sites], "length": length,
                      "out": [[s.start, s.score, s.strand] for s in res[0][0]]})
    return cases


def make_formats(R, genome, GR, toy, chroms):
    """N4: the reference's own data files (data, not code) with what its parsers make of them, which
    malformed files it rejects at which line, and the exact text its writers produce."""
    import glob
    import tempfile
    from motifscan.motif import MotifPfms, MotifPwms
    from motifscan import io as rio, stats as rstats
    n4 = {"files": {}, "bad": {}}
    droot = os.path.join(REF, "tests/data/motifs")
    for rel in ("test/test_pfms.jaspar", "test/test_pwms.motifscan"):
        n4["files"][rel] = open(os.path.join(droot, rel)).read()
    pf = MotifPfms()
    pf.read_pfms(os.path.join(droot, "test/test_pfms.jaspar"))
    n4["pfms"] = [{"matrix_id": p.matrix_id, "name": p.name, "matrix": p.matrix.tolist()} for p in pf]
    pw = MotifPwms()
    pw.read_motifscan_pwms(os.path.join(droot, "test/test_pwms.motifscan"))
    n4["pwms"] = [{"matrix_id": p.matrix_id, "name": p.name, "matrix": p.matrix.tolist(), "cutoffs": p.cutoffs} for p in pw]
    for path in sorted(glob.glob(os.path.join(droot, "bad", "*"))):
        rel = "bad/" + os.path.basename(path)
        n4["files"][rel] = open(path).read()
        try:
            if path.endswith(".jaspar"):
                MotifPfms().read_pfms(path)
            else:
                MotifPwms().read_motifscan_pwms(path)
            n4["bad"][rel] = None
        except Exception as e:                     # noqa: BLE001 - record what the reference says
            n4["bad"][rel] = str(e)
    with tempfile.TemporaryDirectory() as tmp:
        # writer of the built-PWM format, incl. a freshly built PWM (5-decimal log-odds, %8.5f columns)
        built = R["PFM"](n4["pfms"][1]["matrix"], name="Alx1", matrix_id="MA0854.1").to_ppm().to_pwm(
            {"A": 0.3, "C": 0.3, "G": 0.15, "T": 0.25})
        built.set_cutoff("1e-3", 0.20892548)
        built.set_cutoff("1e-4", float(np.around(0.46693615340298805, 8)))
        mp = MotifPwms(list(pw) + [built])
        path = os.path.join(tmp, "w.motifscan")
        mp.write_motifscan_pwms(path)
        n4["written_pwms"] = open(path).read()
        n4["built_matrix"] = built.matrix.tolist()
        # result writers on the whole-chromosome scan of the toy genome at p=1e-2 (both PWMs) and on a
        # second region set with more sites
        whole = [GR(chrom=c, start=0, end=len(s)) for c, s in chroms.items()]
        pwl = list(pw)
        sites = R["scanner"].Scanner(genome=genome, regions=whole, window_size=0, p_value="1e-2").scan_motifs(pwl)
        ctrl_regs = [GR(chrom="chr2", start=0, end=12), GR(chrom="chrX", start=2, end=16), GR(chrom="chrM", start=0, end=15)]
        ctrl = R["scanner"].Scanner(genome=genome, regions=ctrl_regs, window_size=0, p_value="1e-2").scan_motifs(pwl)
        rio.write_sites_table(tmp, pwl, whole, sites)
        rio.write_sites_bed(tmp, pwl, whole, sites)
        enr = rstats.motif_enrichment(pwl, sites, ctrl)
        rio.write_enrich_table(tmp, enr)
        n4["writers"] = {
            "regions": [[r.chrom, r.start, r.end] for r in whole],
            "control_regions": [[r.chrom, r.start, r.end] for r in ctrl_regs],
            "p_value": "1e-2",
            "motif_sites_number.xls": open(os.path.join(tmp, "motif_sites_number.xls")).read(),
            "motif_sites_score.xls": open(os.path.join(tmp, "motif_sites_score.xls")).read(),
            "motif_enrichment.xls": open(os.path.join(tmp, "motif_enrichment.xls")).read(),
            "bed": {os.path.basename(p): open(p).read() for p in sorted(glob.glob(os.path.join(tmp, "motif_sites", "*.bed")))},
        }
    return n4


# -------------------------------------------------------------------------- random --

def random_pfm(rng, width):
    depth = rng.integers(20, 3001)
    cols = rng.dirichlet(0.3 * np.ones(4), size=width).T          # 4 x W
    counts = np.rint(cols * depth).astype(np.int64)
    zero = counts.sum(axis=0) == 0
    counts[:, zero] = 1
    return counts


def build_pwms(R, rng, widths, bg):
    mats = []
    for w in widths:
        pfm = R["PFM"](random_pfm(rng, int(w)))
        mats.append(pfm.to_ppm().to_pwm(bg).matrix)               # matrix.py:74-171
    return mats


def reference_cutoffs(R, mats, rng, bgp, n_kmers, n_threads, batch=64):
    """cli/motif.py:119-153 with n_repeat=1: background k-mers -> c_score(strand 3) ->
    get_score_cutoffs -> around(,8).  K-mers are iid from the background (no genome here)."""
    wmax = max(m.shape[1] for m in mats)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    kmers_arr = letters[rng.choice(4, size=(n_kmers, wmax), p=bgp)]
    kmers = [row.tobytes().decode() for row in kmers_arr]
    all_cut = []
    for i in range(0, len(mats), batch):
        sub = [m.tolist() for m in mats[i:i + batch]]
        scores = R["ext"].c_score(sub, kmers, 3, n_threads)
        for d in R["get_score_cutoffs"](scores):
            all_cut.append({k: float(np.around(np.mean([v]), 8)) for k, v in d.items()})
        del scores
        print(f"  cutoffs {min(i + batch, len(mats))}/{len(mats)}", flush=True)
    return all_cut, kmers_arr


def random_sequences(rng, n, lo, hi, bgp, frac_n=0.02, frac_lower=0.3, iupac=True):
    seqs = []
    for i in range(n):
        L = int(rng.integers(lo, hi + 1))
        a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, size=L, p=bgp)].copy()
        if L and rng.random() < frac_n:
            run = int(rng.integers(1, 51))
            st = int(rng.integers(0, L))
            a[st:st + run] = ord("N")
        if iupac and L and rng.random() < 0.02:
            a[int(rng.integers(0, L))] = ord(rng.choice(list("RYKMSWBDHV")))
        lower = rng.random(L) < frac_lower
        a[lower] = a[lower] + 32
        seqs.append(a.tobytes().decode())
    return seqs


def flat_hits(per_pwm):
    rows = [(p, h[0], h[1], h[2], h[3]) for p, hits in enumerate(per_pwm) for h in hits]
    if not rows:
        z = np.zeros(0)
        return z.astype(np.int32), z.astype(np.int64), z.astype(np.int64), z, z.astype(np.int8)
    a = np.array(rows, dtype=object)
    return (a[:, 0].astype(np.int32), a[:, 1].astype(np.int64), a[:, 2].astype(np.int64),
            a[:, 3].astype(np.float64), a[:, 4].astype(np.int8))


def make_random(R):
    rng = np.random.default_rng(20250310)
    bg = {"A": 0.295, "C": 0.205, "G": 0.205, "T": 0.295}
    bgp = np.array([bg[b] for b in "ACGT"])
    widths = np.concatenate([[5, 30, 6, 17, 29, 8], rng.integers(5, 31, size=42)]).astype(np.int32)
    mats = build_pwms(R, rng, widths, bg)
    cuts, kmers = reference_cutoffs(R, mats, rng, bgp, 20000, 8)
    seqs = random_sequences(rng, 360, 200, 500, bgp)
    seqs += ["", "A", "ACGT", "N" * 40, "acgtn" * 7, "ACGTACGTACGTACGTACGTACGTACGTA"]   # short / ragged
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    ml = [m.tolist() for m in mats]
    save = {"reference_version": np.array(R["version"]),
            "widths": widths,
            "pwm_values": np.concatenate([m.ravel() for m in mats]),
            "cutoff_keys": np.array(sorted(cuts[0].keys())),
            "cutoffs": np.array([[c[k] for k in sorted(cuts[0].keys())] for c in cuts]),
            "seq_bytes": np.frombuffer("".join(seqs).encode(), dtype=np.uint8),
            "seq_offsets": np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)}
    # G3: full hit lists at p=1e-4 (strands 1,2,3) and at p=1e-3 (strand 3, denser)
    for pkey in ("1e-4", "1e-3"):
        c = [d[pkey] for d in cuts]
        for strand in ((1, 2, 3) if pkey == "1e-4" else (3,)):
            hits = R["ext"].c_scan_motif(ml, c, seqs, strand, 4)
            m, s, p, v, d = flat_hits(hits)
            tag = f"scan_p{pkey}_s{strand}"
            save[tag + "_motif"], save[tag + "_seq"], save[tag + "_pos"] = m, s, p
            save[tag + "_score"], save[tag + "_strand"] = v, d
            print(f"  {tag}: {len(m)} hits")
    # G5: c_score on k-mers (first 2000) for the three strand modes
    ksub = [row.tobytes().decode() for row in kmers[:2000]]
    save["kmer_bytes"] = kmers[:2000].copy()
    for strand in (1, 2, 3):
        save[f"score_s{strand}"] = np.array(R["ext"].c_score(ml, ksub, strand, 4))
    # ... and the reference's cutoff pick on those 2000 scores (motif/__init__.py:378-401)
    cuts2000 = R["get_score_cutoffs"](R["ext"].c_score(ml, ksub, 3, 4))
    save["g5_cutoff_keys"] = np.array(sorted(cuts2000[0].keys()))
    save["g5_cutoffs"] = np.array([[c[k] for k in sorted(cuts2000[0].keys())] for c in cuts2000])
    # G4: the real Scanner (window extraction + regroup + de-dup) on a synthetic 3-chromosome genome
    chroms = {f"chr{i + 1}": random_sequences(rng, 1, 20000, 20000, bgp, frac_n=1.0, iupac=False)[0]
              for i in range(3)}
    genome = DictGenome(chroms)
    regs = []
    for i in range(120):
        ch = f"chr{int(rng.integers(1, 4))}"
        st = int(rng.integers(0, 19000))
        en = st + int(rng.integers(30, 900))
        en = min(en, 20000)
        sm = int(rng.integers(st, en))
        regs.append((ch, st, en, sm))
    regs += [("chr1", 0, 50, 10), ("chr2", 19950, 20000, 19990), ("chr3", 5, 12, 7)]     # clipping / short
    gregs = [R["GenomicRegion"](chrom=c, start=s, end=e, summit=m) for c, s, e, m in regs]

    class P:                                   # what scan_motifs reads from a PWM (scanner.py:101-130)
        def __init__(self, m, c):
            self.matrix, self.cutoffs, self.length = m, c, m.shape[1]

    pw = [P(m, c) for m, c in zip(mats, cuts)]
    save["g4_chrom_names"] = np.array(list(chroms.keys()))
    save["g4_chrom_bytes"] = np.frombuffer("".join(chroms.values()).encode(), dtype=np.uint8)
    save["g4_regions"] = np.array([[int(c[3:]) - 1, s, e, m] for c, s, e, m in regs], dtype=np.int64)
    for wsize in (0, 200, 201):
        for dup in (True, False):
            for strand in ("both", "+"):
                if strand == "+" and not (wsize == 200 and dup):
                    continue
                sc_ = R["scanner"].Scanner(genome=genome, regions=gregs, window_size=wsize, strand=strand,
                                           p_value="1e-3", remove_dup=dup)
                rows = sites_to_rows(sc_.scan_motifs(pw))
                tag = f"g4_w{wsize}_dup{int(dup)}_{'both' if strand == 'both' else 'fwd'}"
                save[tag + "_motif"] = np.array([r[0] for r in rows], dtype=np.int32)
                save[tag + "_region"] = np.array([r[1] for r in rows], dtype=np.int64)
                save[tag + "_start"] = np.array([r[2] for r in rows], dtype=np.int64)
                save[tag + "_score"] = np.array([r[3] for r in rows], dtype=np.float64)
                save[tag + "_strand"] = np.array([1 if r[4] == "+" else 2 for r in rows], dtype=np.int8)
                save[tag + "_seq_starts"] = np.array(sc_.seq_starts, dtype=np.int64)
                save[tag + "_seq_ends"] = np.array(sc_.seq_ends, dtype=np.int64)
                print(f"  {tag}: {len(rows)} sites")
    np.savez_compressed(os.path.join(HERE, "ref_random.npz"), **save)
    print("wrote ref_random.npz")


# ---------------------------------------------------------------------------- 579 --

def make_579(R, n_kmers):
    """the synthetic stand-in for JASPAR vertebrates non-redundant (no network)."""
    rng = np.random.default_rng(20250310)
    bg = {"A": 0.295, "C": 0.205, "G": 0.205, "T": 0.295}
    bgp = np.array([bg[b] for b in "ACGT"])
    P = 579
    widths = np.clip(np.rint(rng.gamma(shape=7.5, scale=1.55, size=P)), 5, 30).astype(np.int32)
    widths[0], widths[1] = 30, 5
    mats = build_pwms(R, rng, widths, bg)
    cuts, _ = reference_cutoffs(R, mats, rng, bgp, n_kmers, 8)
    keys = sorted(cuts[0].keys())
    np.savez_compressed(os.path.join(HERE, "..", "..", "motifscan_amd", "data", "synth_jaspar579.npz"),
                        reference_version=np.array(R["version"]), n_kmers=np.array(n_kmers),
                        widths=widths, pwm_values=np.concatenate([m.ravel() for m in mats]),
                        cutoff_keys=np.array(keys),
                        cutoffs=np.array([[c[k] for k in keys] for c in cuts]),
                        bg=bgp)
    print(f"wrote synth_jaspar579.npz  (mean width {widths.mean():.2f}, keys {keys})")


# ------------------------------------------------------------- 579, JASPAR-like information --

def column_with_information(rng, bits):
    """A probability column whose information content (against a uniform background) is `bits`: one dominant base, the rest of
    the mass spread unevenly over the other three (bisection on the dominant base's probability)."""
    w = rng.dirichlet(np.ones(3))                                   # how the rest is split
    lo, hi = 0.25, 1.0 - 1e-9
    for _ in range(60):
        p = 0.5 * (lo + hi)
        q = np.concatenate([[p], (1 - p) * w])
        info = 2.0 + float((q * np.log2(np.maximum(q, 1e-300))).sum())
        lo, hi = (p, hi) if info < bits else (lo, p)
    col = np.concatenate([[p], (1 - p) * w])
    return col[rng.permutation(4)]


def make_579_realistic(R, n_kmers):
    """a second 579-motif set whose per-column information follows a JASPAR-like profile -- an informative core
    (1.2 ... 1.8 bits per column) between low-information flanks (0.1 ... 0.4 bits), 10 % weak motifs (< 6 bits in total) -- built
    through the reference's own to_ppm().to_pwm() (matrix.py:74-171) with cutoffs from its c_score + get_score_cutoffs
    (motif/__init__.py:378-401).  The benchmark set (make_579: Dirichlet(0.3) columns, every column informative) is the one
    BASELINE's numbers are quoted on; this one is a side workload for the filter's candidate ratio and the truncation decision."""
    rng = np.random.default_rng(20261004)
    bg = {"A": 0.295, "C": 0.205, "G": 0.205, "T": 0.295}
    bgp = np.array([bg[b] for b in "ACGT"])
    P = 579
    widths = np.clip(np.rint(rng.gamma(shape=7.5, scale=1.55, size=P)), 5, 30).astype(np.int32)
    widths[0], widths[1] = 30, 5
    mats, infos = [], []
    for w in widths:
        w = int(w)
        depth = int(rng.integers(20, 3001))
        weak = rng.random() < 0.10
        n_core = max(3, int(round(w * rng.uniform(0.45, 0.7))))
        c0 = int(rng.integers(0, w - n_core + 1))
        bits = np.where((np.arange(w) >= c0) & (np.arange(w) < c0 + n_core), rng.uniform(1.2, 1.8, size=w), rng.uniform(0.1, 0.4, size=w))
        if weak:
            bits *= min(1.0, rng.uniform(3.5, 5.9) / bits.sum())
        cols = np.stack([column_with_information(rng, b) for b in bits], axis=1)              # 4 x W
        counts = np.rint(cols * depth).astype(np.int64)
        counts[:, counts.sum(axis=0) == 0] = 1
        mats.append(R["PFM"](counts).to_ppm().to_pwm(bg).matrix)
        infos.append(float(bits.sum()))
    cuts, _ = reference_cutoffs(R, mats, rng, bgp, n_kmers, 8)
    keys = sorted(cuts[0].keys())
    np.savez_compressed(os.path.join(HERE, "..", "..", "motifscan_amd", "data", "synth_jaspar579_lowinfo.npz"),
                        reference_version=np.array(R["version"]), n_kmers=np.array(n_kmers),
                        widths=widths, pwm_values=np.concatenate([m.ravel() for m in mats]),
                        cutoff_keys=np.array(keys),
                        cutoffs=np.array([[c[k] for k in keys] for c in cuts]),
                        bg=bgp, information_bits=np.array(infos))
    print(f"wrote synth_jaspar579_lowinfo.npz  (mean width {widths.mean():.2f}, mean information {np.mean(infos):.1f} bits, "
          f"{int((np.array(infos) < 6).sum())} motifs under 6 bits)")


def make_genome():
    """ref_genome.json (round 6, N3): the reference tests' toy genome as DATA -- tests/data/genomes/test/test.fa and its .fai, byte for
    byte -- with the answers the reference's own tests pin for it (tests/test_genome_class.py:14-23: sorted chromosome names, sizes, whole-
    chromosome fetches; tests/test_scanner.py:17,22: chr1[2:5] = 'TtC', chr1[1:5] = 'aTtC'), every (start, end) slice of every chromosome
    (pysam's fetch of an in-range request is the 0-based half-open slice, which those literals pin), and two re-wrapped copies of the same
    records (7 bases per line; CRLF line ends with a description behind the name) that must read back the same."""
    d = os.path.join(REF, "tests/data/genomes/test")
    fa = open(os.path.join(d, "test.fa")).read()
    fai = open(os.path.join(d, "test.fa.fai")).read()
    chroms = read_fasta(os.path.join(d, "test.fa"))
    # the literals of tests/test_genome_class.py:14-23
    assert sorted(chroms) == ["chr1", "chr2", "chrM", "chrX"]
    assert {k: len(v) for k, v in chroms.items()} == {"chr1": 10, "chr2": 17, "chrM": 15, "chrX": 16}
    assert chroms["chr1"] == "AaTtCcGgNn" and chroms["chr2"] == "AAAaCCccTTtGNNNNN" and chroms["chrM"] == "AaaaaAAAAAAAnnn" and chroms["chrX"] == "AAACCTACNNTnggAC"
    assert chroms["chr1"][2:5] == "TtC" and chroms["chr1"][1:5] == "aTtC"        # tests/test_scanner.py:17,22
    wrapped = "".join(f">{k}\n" + "".join(v[i:i + 7] + "\n" for i in range(0, len(v), 7)) for k, v in chroms.items())
    crlf = "".join(f">{k} some description\r\n" + "".join(v[i:i + 5] + "\r\n" for i in range(0, len(v), 5)) + "\r\n" for k, v in chroms.items())
    out = {"files": {"test.fa": fa, "test.fa.fai": fai, "wrapped7.fa": wrapped, "crlf5.fa": crlf},
           "order": list(chroms), "chroms_sorted": sorted(chroms), "chrom_sizes": {k: len(v) for k, v in chroms.items()},
           "whole": dict(chroms),
           "reference_test_fetches": [["chr1", 0, 10, "AaTtCcGgNn"], ["chr2", 0, 17, "AAAaCCccTTtGNNNNN"], ["chrM", 0, 15, "AaaaaAAAAAAAnnn"],
                                      ["chrX", 0, 16, "AAACCTACNNTnggAC"], ["chr1", 2, 5, "TtC"], ["chr1", 1, 5, "aTtC"]],
           "all_slices": {k: [[a, b, v[a:b]] for a in range(len(v) + 1) for b in range(a, len(v) + 1)] for k, v in chroms.items()}}
    with open(os.path.join(HERE, "ref_genome.json"), "w") as fh:
        json.dump(out, fh, indent=0)
    print("wrote ref_genome.json")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only-genome", action="store_true", help="only ref_genome.json (the toy genome's data files and the reference tests' answers for them)")
    ap.add_argument("--skip-579", action="store_true")
    ap.add_argument("--only-579", action="store_true")
    ap.add_argument("--only-lowinfo", action="store_true", help="only the JASPAR-like-information 579-motif side set")
    ap.add_argument("--only-dedup", action="store_true", help="refresh only the de-dup cases of ref_small.json (everything else stays byte for byte)")
    ap.add_argument("--n-kmers", type=int, default=1000000)
    a = ap.parse_args()
    if a.only_genome:
        make_genome()
        sys.exit(0)
    R = import_reference()
    if a.only_lowinfo:
        make_579_realistic(R, a.n_kmers)
        sys.exit(0)
    if a.only_dedup:
        path = os.path.join(HERE, "ref_small.json")
        with open(path) as fh:
            small = json.load(fh)
        small["dedup"] = dedup_cases(R["scanner"])
        with open(path, "w") as fh:
            json.dump(small, fh, indent=1)
        print("refreshed the de-dup cases of ref_small.json:", len(small["dedup"]))
        sys.exit(0)
    if not a.only_579:
        make_small(R)
        make_random(R)
        make_genome()
    if not a.skip_579:
        make_579(R, a.n_kmers)
