"""
Pins the oracle (oracle/cscore_oracle.c + oracle/oracle.py) against
  * the literal known answers in the reference's own tests, and
  * the golden vectors the real reference produced (tests/golden/make_golden.py).
CPU only.  Everything is compared bit-for-bit unless the reference's own test uses approx.
"""
import numpy as np
import pytest

M3 = [[[1.35, 0.21, -5.23], [0.07, -0.21, 0.6], [2.15, 2.22, -0.84], [-2.64, -1.89, 5.47]]]


# -- literal expectations copied from the reference's tests (they are data: inputs + expected outputs) --

def test_known_answer_c_score(oracle):
    """/root/reference/tests/test_motif_score.py:6-20"""
    seqs = ["NNN", "AGT", "ANT", "CTA"]
    assert oracle.c_score(M3, seqs, 1, 1)[0] == pytest.approx(
        [0.0, 0.9186991869918698, 0.693089430894309, -0.7164634146341464])
    assert oracle.c_score(M3, seqs, 2, 1)[0] == pytest.approx(
        [0.0, 0.6717479674796748, 0.693089430894309, -0.3323170731707317])
    assert oracle.c_score(M3, seqs, 3, 1)[0] == pytest.approx(
        [0.0, 0.9186991869918698, 0.693089430894309, -0.3323170731707317])


def test_known_answer_c_scan_motif(oracle):
    """/root/reference/tests/test_motif_score.py:23-32 (values and ORDER)"""
    sites = oracle.c_scan_motif(M3, [0.2], ["NNNAG", "TANTCTA"], 3, 1)
    assert len(sites) == 1 and len(sites[0]) == 4
    assert sites[0][0] == pytest.approx([1, 1, 0.693089430894309, 1])
    assert sites[0][1] == pytest.approx([1, 1, 0.693089430894309, 2])
    assert sites[0][2] == pytest.approx([1, 2, 0.23983739837398374, 2])
    assert sites[0][3] == pytest.approx([1, 3, 0.266260162601626, 1])


def test_known_answer_dedup(oracle):
    """/root/reference/tests/test_scanner.py:57-73"""
    S = oracle.MotifSite
    sites = [S(1, 1, "+"), S(3, 0.8, "+"), S(1, 1, "-"), S(2, 3, "-"), S(5, 1, "+")]
    out = oracle.deduplicate_motif_sites([[sites]], [3])
    assert [(s.start, s.strand) for s in out[0][0]] == [(1, "+"), (2, "-"), (5, "+")]


# -- golden vectors from the real reference ----------------------------------------------

def test_g1_exact(oracle, small):
    g = small["G1"]
    for s in ("1", "2", "3"):
        assert oracle.c_score(g["matrix"], g["score_seqs"], int(s), 1) == g["score"][s]
        assert oracle.c_scan_motif(g["matrix"], g["scan_cutoffs"], g["scan_seqs"], int(s), 1) == g["scan"][s]


def test_g6_edge_cases_exact(oracle, small):
    for case in small["G6"]:
        if case["kind"] == "scan":
            got = oracle.c_scan_motif(case["pwms"], case["cutoffs"], case["seqs"], case["strand"], 1)
        else:
            got = oracle.c_score(case["pwms"], case["seqs"], case["strand"], 1)
        assert got == case["out"], case["name"]


def test_dedup_cases(oracle, small):
    S = oracle.MotifSite
    for case in small["dedup"]:
        sites = [S(*s) for s in case["sites"]]
        out = oracle.deduplicate_motif_sites([[sites]], [case["length"]])
        assert [[s.start, s.score, s.strand] for s in out[0][0]] == case["out"], case["name"]


@pytest.mark.parametrize("tag,strand", [("scan_p1e-4_s1", 1), ("scan_p1e-4_s2", 2), ("scan_p1e-4_s3", 3),
                                        ("scan_p1e-3_s3", 3)])
@pytest.mark.parametrize("n_threads", [1, 4])
def test_g3_random_scan_exact(oracle, rnd, tag, strand, n_threads):
    pkey = tag.split("_")[1][1:]
    vals, widths = oracle.flatten_pwms(rnd["mats"])
    bases, offsets = oracle.flatten_seqs(rnd["seqs"])
    r = oracle.scan_arrays(vals, widths, rnd["cutoff_by_key"][pkey], bases, offsets, strand, n_threads)
    motif = np.repeat(np.arange(len(widths)), np.diff(r["motif_offsets"]))
    assert np.array_equal(motif, rnd[tag + "_motif"])
    assert np.array_equal(r["seq_idx"], rnd[tag + "_seq"])
    assert np.array_equal(r["pos"], rnd[tag + "_pos"])
    assert np.array_equal(r["strand"], rnd[tag + "_strand"])
    assert np.array_equal(r["score"], rnd[tag + "_score"])          # bit-exact fp64


def test_g5_kmer_scores_exact(oracle, rnd):
    kmers = [row.tobytes().decode() for row in rnd["kmer_bytes"]]
    vals, widths = oracle.flatten_pwms(rnd["mats"])
    bases, offsets = oracle.flatten_seqs(kmers)
    for strand in (1, 2, 3):
        got = oracle.score_arrays(vals, widths, bases, offsets, strand, 2)
        assert np.array_equal(got, rnd[f"score_s{strand}"])


def test_g4_scanner_postprocessing(oracle, rnd):
    """scan -> make_motif_sites -> de-dup restated in oracle.py == the real Scanner.scan_motifs."""
    names = [str(x) for x in rnd["g4_chrom_names"]]
    raw = rnd["g4_chrom_bytes"].tobytes().decode()
    n = len(raw) // len(names)
    chroms = [raw[i * n:(i + 1) * n] for i in range(len(names))]
    regs = rnd["g4_regions"]
    cut = rnd["cutoff_by_key"]["1e-3"]
    lengths = [m.shape[1] for m in rnd["mats"]]
    for wsize, dup, strand in ((0, True, 3), (0, False, 3), (200, True, 3), (200, True, 1), (201, False, 3)):
        tag = f"g4_w{wsize}_dup{int(dup)}_{'both' if strand == 3 else 'fwd'}"
        starts, ends = rnd[tag + "_seq_starts"], rnd[tag + "_seq_ends"]
        seqs = [chroms[int(c)][int(s):int(e)] for (c, _, _, _), s, e in zip(regs, starts, ends)]
        sites = oracle.c_scan_motif([m.tolist() for m in rnd["mats"]], cut.tolist(), seqs, strand, 4)
        ms = oracle.make_motif_sites(sites, [int(s) for s in starts])
        if dup:
            ms = oracle.deduplicate_motif_sites(ms, lengths)
        rows = [(p, r, s.start, s.score, 1 if s.strand == "+" else 2)
                for p, per in enumerate(ms) for r, ss in enumerate(per) for s in ss]
        assert [x[0] for x in rows] == rnd[tag + "_motif"].tolist()
        assert [x[1] for x in rows] == rnd[tag + "_region"].tolist()
        assert [x[2] for x in rows] == rnd[tag + "_start"].tolist()
        assert [x[4] for x in rows] == rnd[tag + "_strand"].tolist()
        assert np.array_equal(np.array([x[3] for x in rows]), rnd[tag + "_score"])


def test_max_raw_score_clamps_at_zero(oracle, small):
    """cscore.c:39: column max starts at 0 (differs from numpy's matrix.max(0).sum())."""
    assert oracle.max_raw_score([[-1, 2], [-2, 1], [-3, .5], [-4, .1]]) == 2.0
    assert oracle.max_raw_score(small["G7"]["pwm"]) == pytest.approx(9.84)


def test_live_reference_agrees_when_present(oracle, rnd):
    """In the build container the real extension is in oracle/_ref: compare a fresh random case."""
    ref = oracle.load_reference_ext()
    if ref is None:
        pytest.skip("oracle/_ref not built on this box")
    rng = np.random.default_rng(7)
    mats = [np.round(rng.normal(0, 1.5, size=(4, int(w))), 5) for w in rng.integers(4, 34, size=12)]
    seqs = ["".join(rng.choice(list("ACGTNacgt"), size=int(n))) for n in rng.integers(0, 120, size=60)]
    cut = rng.uniform(-0.2, 0.6, size=len(mats)).tolist()
    ml = [m.tolist() for m in mats]
    for strand in (1, 2, 3):
        assert oracle.c_scan_motif(ml, cut, seqs, strand, 3) == ref.c_scan_motif(ml, cut, seqs, strand, 3)
    long_seqs = [s for s in seqs if len(s) >= 34]
    assert oracle.c_score(ml, long_seqs, 3, 2) == ref.c_score(ml, long_seqs, 3, 2)


def test_live_reference_agrees_on_decision_boundary_fuzz(oracle):
    """tests/fuzz_parity.py's cases (ties, cutoffs on attainable scores +- 1 ulp / 1e-10, max_raw == 0,
    huge and tiny magnitudes) through the oracle and through the real extension."""
    ref = oracle.load_reference_ext()
    if ref is None:
        pytest.skip("oracle/_ref not built on this box")
    import fuzz_parity
    n_hits = 0
    for seed in range(60):
        mats, cutoffs, seqs, strand = fuzz_parity.make_case(seed)
        ml = [m.tolist() for m in mats]
        got = oracle.c_scan_motif(ml, cutoffs.tolist(), seqs, strand, 2)
        assert got == ref.c_scan_motif(ml, cutoffs.tolist(), seqs, strand, 2), seed
        n_hits += sum(len(g) for g in got)
    assert n_hits > 10000
