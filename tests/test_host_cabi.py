"""
CPU-only checks of the host side of libmotifscan_amd.so: the library loads and exports every
symbol the headers declare, argument validation maps to the reference's exception classes, the
pre-filter quantiser can never drop a window the reference reports, and the host-side
de-duplication equals the reference's.  No compute entry point is called (no GPU here).
"""
import ctypes
import os
import re

import numpy as np
import pytest

from motifscan_amd import _lib, matrix, scanner

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(_lib.LIB_PATH)
    declared = set()
    for hdr in ("motifscan_amd.h", "motifscan_amd_debug.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(ms_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/ but not exported"


def test_scan_stats_struct_matches_header():
    text = open(os.path.join(ROOT, "include", "motifscan_amd.h")).read()
    body = re.search(r"typedef struct ms_scan_stats \{(.*?)\} ms_scan_stats;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(int64_t|int32_t|double)\s+(\w+);", body)
    ctmap = {"int64_t": ctypes.c_int64, "int32_t": ctypes.c_int32, "double": ctypes.c_double}
    assert [(n, ctmap[t]) for t, n in fields] == list(_lib.ScanStats._fields_)


def test_validation_maps_to_reference_exceptions():
    with pytest.raises(ValueError):
        _lib.PwmSet.from_matrices([[[1.0], [2.0], [3.0]]])                 # 3 rows
    with pytest.raises(ValueError):
        _lib.PwmSet.from_matrices([[[], [], [], []]])                      # width 0
    with pytest.raises(ValueError):
        _lib.PwmSet(np.zeros(7), np.array([2], dtype=np.int32))            # wrong value count
    with pytest.raises(ValueError):
        _lib.PwmSet.from_matrices([np.zeros((4, 3))], cutoffs=[0.1, 0.2])


def test_no_gpu_means_loud_failure_not_fallback():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError):
        _lib.SeqSet.from_strings(["ACGT"])
    from motifscan_amd import cscore
    with pytest.raises(RuntimeError):
        cscore.c_scan_motif([np.zeros((4, 3)).tolist()], [0.5], ["ACGTACGT"], 3, 1)


def test_max_raw_is_the_c_definition(oracle, rnd):
    pw = _lib.PwmSet.from_matrices(rnd["mats"] + [np.array([[-1, 2], [-2, 1], [-3, .5], [-4, .1]])])
    got = pw.max_raw()
    want = [oracle.max_raw_score(m) for m in rnd["mats"]] + [2.0]
    assert np.array_equal(got, np.array(want))


# ---------------------------------------------------------------- pre-filter plan --

def emulate_prefilter(plan, seq_codes_2bit):
    """numpy model of prefilter_kernel: for every window start of one N-free sequence return the
    set of (motif, pos, strand) whose field reaches its top bit."""
    L = len(seq_codes_2bit)
    padded = np.concatenate([seq_codes_2bit, np.zeros(40, dtype=np.int64)])
    code = padded[:-1] | (padded[1:] << 2)                    # 2-mer code at every position
    flagged = set()
    for q in range(plan["group_motifs"].shape[0]):
        G, fb = int(plan["group_G"][q]), int(plan["group_fb"][q])
        nf = 32 // fb
        acc = np.zeros((L, 4), dtype=np.uint64)
        for g in range(G):
            acc += plan["tables"][q, g][code[2 * g:2 * g + L]]
        assert (acc < (1 << 32)).all()                        # the kernel adds in 32 bits
        for n in range(4 * nf):                               # field n: word n & 3, field n >> 2
            field = (acc[:, n & 3] >> np.uint64((n >> 2) * fb)) & np.uint64((1 << fb) - 1)
            m = int(plan["group_motifs"][q, n >> 1])
            hot = np.nonzero(field >= (1 << (fb - 1)))[0]
            if m < 0:
                assert len(hot) == 0
                continue
            for j in hot:
                flagged.add((m, int(j), 1 + (n & 1)))
        if nf * fb < 32:                                      # unused high bits of every word stay clear
            assert not (acc >> np.uint64(nf * fb)).any()
    return flagged


def fields_never_overflow(plan):
    """Worst case of every field (sum over 2-mer positions of the largest entry) fits its width, so
    no carry can ever cross into the neighbouring field."""
    t = plan["tables"].astype(np.uint64)
    for q in range(t.shape[0]):
        fb = int(plan["group_fb"][q])
        for n in range(4 * (32 // fb)):
            field = (t[q, :, :, n & 3] >> np.uint64((n >> 2) * fb)) & np.uint64((1 << fb) - 1)
            assert int(field.max(axis=1).sum()) <= (1 << fb) - 1


@pytest.mark.parametrize("pkey", ["1e-2", "1e-3", "1e-4"])
@pytest.mark.parametrize("strand", [1, 2, 3])
def test_prefilter_never_loses_a_reference_hit(oracle, rnd, pkey, strand, monkeypatch):
    monkeypatch.setenv("MS_MEASURE", "1")            # measurement / A-B switches need the explicit opt-in
    monkeypatch.setenv("MS_PF_ENGINE", "0")
    mats, cut = rnd["mats"], rnd["cutoff_by_key"][pkey]
    pw = _lib.PwmSet.from_matrices(mats, cut)
    plan = pw.plan(strand)
    fields_never_overflow(plan)
    assert plan["n_fast"] + plan["n_exact"] == len(mats)
    fast = set(plan["group_motifs"].ravel().tolist()) - {-1}
    assert fast | set(plan["exact_motifs"].tolist()) == set(range(len(mats)))
    rng = np.random.default_rng(5)
    seqs = ["".join(rng.choice(list("ACGT"), p=[.295, .205, .205, .295], size=3000)) for _ in range(3)]
    lut = {c: i for i, c in enumerate("ACGT")}
    n_flag = n_hit = 0
    for s in seqs:
        codes = np.array([lut[c] for c in s], dtype=np.int64)
        flagged = emulate_prefilter(plan, codes)
        sites = oracle.c_scan_motif([m.tolist() for m in mats], cut.tolist(), [s], strand, 2)
        for m, hits in enumerate(sites):
            if m not in fast:
                continue
            for _, pos, _, sd in hits:
                assert (m, pos, sd) in flagged, (m, pos, sd)
                n_hit += 1
        n_flag += sum(1 for (m, j, sd) in flagged if j + mats[m].shape[1] <= len(s))
    # the filter must stay selective: allow 2.5x the true hits plus slack
    assert n_flag <= 2.5 * n_hit + 200, (n_flag, n_hit)


def emulate_mfma_prefilter(plan, seq_codes_2bit):
    """numpy model of prefilter_mfma_kernel: i32 sums of int8 rows over the one-hot sequence image;
    a (field, window) is a candidate iff the sum is >= 0."""
    L = len(seq_codes_2bit)
    padded = np.concatenate([seq_codes_2bit, np.zeros(40, dtype=np.int64)])
    flagged = set()
    rows = plan["rows"].astype(np.int64)
    for q in range(rows.shape[0]):
        ncol = min(32, plan["cols_per_kb"] * int(plan["group_kb"][q]))
        assert not rows[q, :, ncol:, :].any()
        for n in range(16):
            m = int(plan["group_motifs"][q, n >> 1])
            acc = np.full(L, int(plan["bias"][q, n]), dtype=np.int64)
            for c in range(ncol):
                acc += rows[q, n, c][padded[c:c + L]]
            hot = np.nonzero(acc >= 0)[0]
            if m < 0:
                assert len(hot) == 0                          # empty slots never flag
                continue
            for j in hot:
                flagged.add((m, int(j), 1 + (n & 1)))
    return flagged


@pytest.mark.parametrize("engine", ["1", "2", "3"])
@pytest.mark.parametrize("pkey", ["1e-2", "1e-3", "1e-4"])
@pytest.mark.parametrize("strand", [1, 2, 3])
def test_mfma_prefilter_never_loses_a_reference_hit(oracle, rnd, pkey, strand, monkeypatch, engine):
    monkeypatch.setenv("MS_MEASURE", "1")            # measurement / A-B switches need the explicit opt-in
    monkeypatch.setenv("MS_PF_ENGINE", engine)
    mats, cut = rnd["mats"], rnd["cutoff_by_key"][pkey]
    pw = _lib.PwmSet.from_matrices(mats, cut)
    plan = pw.plan_mfma(strand)
    assert plan["n_fast"] + plan["n_exact"] == len(mats)
    fast = set(plan["group_motifs"].ravel().tolist()) - {-1}
    assert fast | set(plan["exact_motifs"].tolist()) == set(range(len(mats)))
    assert plan["rows"].shape[0] % 2 == 0                     # two table groups per 32-row operand tile
    rng = np.random.default_rng(5)
    seqs = ["".join(rng.choice(list("ACGT"), p=[.295, .205, .205, .295], size=3000)) for _ in range(3)]
    lut = {c: i for i, c in enumerate("ACGT")}
    n_flag = n_hit = 0
    for s in seqs:
        codes = np.array([lut[c] for c in s], dtype=np.int64)
        flagged = emulate_mfma_prefilter(plan, codes)
        assert all((strand >> (sd - 1)) & 1 for (_, _, sd) in flagged)      # a strand not asked for never flags
        sites = oracle.c_scan_motif([m.tolist() for m in mats], cut.tolist(), [s], strand, 2)
        for m, hits in enumerate(sites):
            if m not in fast:
                continue
            for _, pos, _, sd in hits:
                assert (m, pos, sd) in flagged, (m, pos, sd)
                n_hit += 1
        n_flag += sum(1 for (m, j, sd) in flagged if j + mats[m].shape[1] <= len(s))
    # the filter must stay selective (the fp6 grid of engine 3 is coarser: ~56 levels against 127-254, which shows at p = 1e-2)
    assert n_flag <= (3.0 if engine == "3" else 2.0) * n_hit + 200, (n_flag, n_hit)
    if engine == "3":                                         # fp6 e2m3: every entry is on the grid (units of 1/8), rows are exact in f32
        mag = np.abs(plan["rows"].astype(np.int64))
        assert ((mag <= 16) | ((mag <= 32) & (mag % 2 == 0)) | ((mag <= 60) & (mag % 4 == 0))).all()


@pytest.mark.parametrize("engine", ["1", "2", "3"])
def test_mfma_plan_on_decision_boundary_cases(oracle, monkeypatch, engine):
    """The fuzzer's tie-heavy cases (cutoffs exactly on attainable scores): the int8 plans keep every hit."""
    import fuzz_parity
    monkeypatch.setenv("MS_MEASURE", "1")            # measurement / A-B switches need the explicit opt-in
    monkeypatch.setenv("MS_PF_ENGINE", engine)
    lut = {c: i for i, c in enumerate("ACGT")}
    checked = 0
    for seed in range(40):
        mats, cutoffs, seqs, strand = fuzz_parity.make_case(seed)
        if len(mats) > 40:
            continue
        seqs = [s.upper() for s in seqs if len(s) >= 8 and set(s.upper()) <= set("ACGT")][:6]
        if not seqs:
            continue
        pw = _lib.PwmSet.from_matrices(mats, cutoffs)
        plan = pw.plan_mfma(strand)
        fast = set(plan["group_motifs"].ravel().tolist()) - {-1}
        sites = oracle.c_scan_motif([m.tolist() for m in mats], cutoffs.tolist(), seqs, strand, 2)
        flagged = [emulate_mfma_prefilter(plan, np.array([lut[c] for c in s], dtype=np.int64)) for s in seqs]
        for m, hits in enumerate(sites):
            if m in fast:
                for si, pos, _, sd in hits:
                    assert (m, pos, sd) in flagged[si], (seed, m, pos, sd)
                    checked += 1
    assert checked > 2000


@pytest.mark.parametrize("engine", ["0", "1"])
def test_prefilter_routes_degenerate_pwms_to_exact_path(monkeypatch, engine):
    monkeypatch.setenv("MS_MEASURE", "1")            # measurement / A-B switches need the explicit opt-in
    monkeypatch.setenv("MS_PF_ENGINE", engine)
    wide = np.zeros((4, 40))
    wide[0] = 1.0
    allneg = -np.ones((4, 5))
    nonfinite = np.ones((4, 6))
    nonfinite[2, 3] = -np.inf
    ok = np.array([[1.0, -2, 0.5], [-1, 1.2, -0.3], [0.2, -0.4, 0.9], [-3, 0.1, -1.0]])
    low_cut = ok.copy()
    pw = _lib.PwmSet.from_matrices([wide, allneg, nonfinite, ok, low_cut], [0.5, 0.5, 0.5, 0.6, -50.0])
    plan = pw.plan(3) if engine == "0" else pw.plan_mfma(3)
    assert sorted(plan["exact_motifs"].tolist()) == [0, 1, 2, 4]
    assert plan["n_fast"] == 1


def test_plan_tiles_respect_lds_budget(jaspar579, monkeypatch):
    monkeypatch.setenv("MS_MEASURE", "1")            # measurement / A-B switches need the explicit opt-in
    monkeypatch.setenv("MS_PF_ENGINE", "0")
    pw = _lib.PwmSet(jaspar579["pwm_values"], jaspar579["widths"], jaspar579["cutoffs"]["1e-4"])
    for budget in (64 * 1024, 143 * 1024):
        plan = pw.plan(3, budget)
        assert plan["n_exact"] == 0 and plan["n_fast"] == 579
        tf = plan["tile_first_group"]
        for t in range(len(tf) - 1):
            tile_bytes = int(plan["group_G"][tf[t]:tf[t + 1]].sum()) * 256
            assert 0 < tile_bytes <= budget
        assert tf[-1] == len(plan["group_G"])
        per_group = np.where(plan["group_fb"] == 10, 6, 4)
        assert (plan["group_motifs"] >= 0).sum() == 579 and ((plan["group_motifs"] >= 0).sum(axis=1) <= per_group).all()
        key = plan["group_fb"].astype(np.int64) * 100 + plan["group_G"]
        assert (np.diff(key) >= 0).all()                      # same field width together, narrow to wide
        assert (plan["group_fb"] == 10).sum() > 0.8 * len(key)   # JASPAR-like motifs mostly take 10-bit fields


def test_mfma_plan_tiles_respect_lds_budget(jaspar579, monkeypatch):
    """Engine 1: 16 motifs x {fwd, rev} per 32-row operand tile, ceil(W_max / 8) KiB each, narrow to wide;
    LDS tiles hold whole row tiles and stay inside the budget."""
    monkeypatch.setenv("MS_MEASURE", "1")            # measurement / A-B switches need the explicit opt-in
    monkeypatch.setenv("MS_PF_ENGINE", "1")
    pw = _lib.PwmSet(jaspar579["pwm_values"], jaspar579["widths"], jaspar579["cutoffs"]["1e-4"])
    widths = jaspar579["widths"]
    monkeypatch.setenv("MS_MEASURE", "1")            # measurement / A-B switches need the explicit opt-in
    monkeypatch.setenv("MS_PF_ENGINE", "2")                    # Walsh form: 10 columns per k-block, fewer k-blocks in total
    p2 = pw.plan_mfma(3, 143 * 1024)
    assert p2["n_fast"] == 579 and int(p2["group_kb"][0::2].sum()) == 62 and p2["group_kb"].max() == 3
    for q in range(len(p2["group_kb"])):
        ws = widths[p2["group_motifs"][q][p2["group_motifs"][q] >= 0]]
        assert len(ws) == 0 or (ws <= 10 * p2["group_kb"][q]).all()
    monkeypatch.setenv("MS_PF_ENGINE", "3")                    # fp6 x fp4: 16 columns per k-block of 1.5 KiB
    p3 = pw.plan_mfma(3, 143 * 1024)
    assert p3["n_fast"] == 579 and int(p3["group_kb"][0::2].sum()) == 43 and p3["group_kb"].max() == 2 and p3["n_tiles"] == 1
    for q in range(len(p3["group_kb"])):
        ws = widths[p3["group_motifs"][q][p3["group_motifs"][q] >= 0]]
        assert len(ws) == 0 or (ws <= 16 * p3["group_kb"][q]).all()
    monkeypatch.setenv("MS_MEASURE", "1")            # measurement / A-B switches need the explicit opt-in
    monkeypatch.setenv("MS_PF_ENGINE", "1")
    for budget in (32 * 1024, 143 * 1024):
        plan = pw.plan_mfma(3, budget)
        assert plan["n_exact"] == 0 and plan["n_fast"] == 579
        gm, kb, tf = plan["group_motifs"], plan["group_kb"], plan["tile_first_group"]
        assert len(kb) == 2 * ((579 + 15) // 16) and (kb[0::2] == kb[1::2]).all()
        assert (np.diff(kb) >= 0).all()
        for q in range(len(kb)):
            ws = widths[gm[q][gm[q] >= 0]]
            assert len(ws) == 0 or (ws <= 8 * kb[q]).all()
        assert sorted(gm[gm >= 0].tolist()) == list(range(579))
        assert plan["n_tiles"] == len(tf) - 1 and tf[-1] == len(kb) and (np.array(tf) % 2 == 0).all()
        for t in range(len(tf) - 1):
            tile_bytes = int(kb[tf[t]:tf[t + 1]:2].sum()) * 1024
            assert 0 < tile_bytes <= budget
        assert (plan["n_tiles"] == 1) == (budget > 100 * 1024)


def test_field_width_switch(jaspar579, monkeypatch):
    monkeypatch.setenv("MS_MEASURE", "1")            # measurement / A-B switches need the explicit opt-in
    monkeypatch.setenv("MS_PF_ENGINE", "0")
    pw = _lib.PwmSet(jaspar579["pwm_values"], jaspar579["widths"], jaspar579["cutoffs"]["1e-4"])
    monkeypatch.setenv("MS_PF_FIELD_BITS", "16")
    plan = pw.plan(3)
    assert (plan["group_fb"] == 16).all() and len(plan["group_G"]) == (579 + 3) // 4
    fields_never_overflow(plan)


# --------------------------------------------------------------------------- dedup --

def test_dedup_matches_reference_cases(small):
    S = scanner.MotifSite
    for case in small["dedup"]:
        sites = [S(*s) for s in case["sites"]]
        out = scanner.deduplicate_motif_sites([[sites]], [case["length"]])
        assert [[s.start, s.score, s.strand] for s in out[0][0]] == case["out"], case["name"]


def test_dedup_matches_oracle_on_random_hits(oracle, rnd):
    tag = "scan_p1e-3_s3"
    motif, seq, pos = rnd[tag + "_motif"], rnd[tag + "_seq"], rnd[tag + "_pos"]
    score, strand = rnd[tag + "_score"], rnd[tag + "_strand"]
    P = len(rnd["widths"])
    offsets = np.concatenate([[0], np.cumsum(np.bincount(motif, minlength=P))])
    keep = _lib.dedup_keep(offsets, rnd["widths"], seq, pos, score, strand)
    sites = [[] for _ in range(P)]
    for m, s, p_, v, d in zip(motif, seq, pos, score, strand):
        sites[m].append([int(s), int(p_), float(v), int(d)])
    ms = oracle.make_motif_sites(sites, [0] * len(rnd["seqs"]))
    dd = oracle.deduplicate_motif_sites(ms, [int(w) for w in rnd["widths"]])
    want = [(m, r, s.start, s.score, 1 if s.strand == "+" else 2)
            for m, per in enumerate(dd) for r, ss in enumerate(per) for s in ss]
    got = list(zip(motif[keep].tolist(), seq[keep].tolist(), pos[keep].tolist(), score[keep].tolist(),
                   strand[keep].tolist()))
    assert got == want
    assert 0 < keep.sum() < len(keep)


def test_cutoff_rank_semantics_match_reference(oracle, rnd):
    """build.cutoff_ranks restates get_score_cutoffs' index arithmetic (motif/__init__.py:393-399):
    picking those ranks from the oracle's scores reproduces the reference's cutoffs bit for bit."""
    from motifscan_amd import build
    assert build.cutoff_ranks(2000) == {"1e-2": 19, "1e-3": 1}
    assert build.cutoff_ranks(1000000) == {"1e-2": 9999, "1e-3": 999, "1e-4": 99, "1e-5": 9, "1e-6": 0}
    assert build.cutoff_ranks(20000) == {"1e-2": 199, "1e-3": 19, "1e-4": 1}
    with pytest.raises(ValueError):
        build.cutoff_ranks(99)
    scores = -np.sort(-rnd["score_s3"], axis=1)                       # descending
    keys = [str(k) for k in rnd["g5_cutoff_keys"]]
    ranks = build.cutoff_ranks(scores.shape[1])
    assert list(ranks) == keys
    assert np.array_equal(scores[:, [ranks[k] for k in keys]], rnd["g5_cutoffs"])


# -------------------------------------------------------------------------- matrix --

def test_matrix_pipeline_matches_reference_values(small):
    g = small["G7"]
    pfm = matrix.PositionFrequencyMatrix(g["pfm"])
    assert np.array_equal(pfm.to_ppm(normalize=False).matrix, np.array(g["pfm_to_ppm_raw"]))
    assert np.array_equal(pfm.to_ppm(normalize=True, pseudo=0.001).matrix, np.array(g["pfm_to_ppm_norm"]))
    ppm = matrix.PositionProbabilityMatrix(g["ppm"])
    ppm.normalize(pseudo=0.001)
    assert np.array_equal(ppm.matrix, np.array(g["ppm_normalized"]))
    assert np.array_equal(ppm.to_pwm().matrix, np.array(g["ppm_to_pwm_default_bg"]))
    assert np.array_equal(ppm.to_pwm(bg_freq=g["bg"]).matrix, np.array(g["ppm_to_pwm_bg"]))
    pwm = matrix.PositionWeightMatrix(g["pwm"])
    assert float(pwm.max_raw_score) == g["max_raw_score"]
    assert float(pwm.min_raw_score) == g["min_raw_score"]
    for s, v in g["score"].items():
        assert float(pwm.score(s)) == v


def test_matrix_errors_like_reference():
    """/root/reference/tests/test_motif_matrix.py:9-60,106-112"""
    with pytest.raises(ValueError):
        matrix.PositionMatrix([[1], [2], [3]])
    with pytest.raises(ValueError):
        matrix.PositionMatrix([[], [], [], []])
    with pytest.raises(ValueError):
        matrix.PositionFrequencyMatrix([[1], [0.4], [7], [10]])
    with pytest.raises(ValueError):
        matrix.PositionFrequencyMatrix([[-1], [4], [7], [10]])
    with pytest.raises(ValueError):
        matrix.PositionFrequencyMatrix([[0], [0], [0], [0]])
    with pytest.raises(ValueError):
        matrix.PositionProbabilityMatrix([[0], [0.2], [-0.1], [0.9]])
    with pytest.raises(ValueError):
        matrix.PositionProbabilityMatrix([[0.3], [0.2], [0.5], [0.3]])
    ppm = matrix.PositionProbabilityMatrix([[0.2, 0.2], [0.2, 0.2], [0.3, 0.6], [0.3, 0]])
    with pytest.raises(ValueError):
        ppm.normalize(pseudo=1)
    pwm = matrix.PositionWeightMatrix([[1.35, 0.21, -5.23], [0.07, -0.21, 0.6], [2.15, 2.22, -0.84],
                                       [-2.64, -1.89, 5.47]])
    with pytest.raises(ValueError):
        pwm.score("")
    with pytest.raises(ValueError):
        pwm.score("NNNN")
    assert pwm.score("NNN") == 0


def test_scanner_ctor_semantics_without_gpu(small):
    """Window extraction is host logic (scanner.py:44-87): check it against the reference's run."""
    g = small["G2"]

    class G:
        chrom_sizes = {k: len(v) for k, v in g["chroms"].items()}

        @staticmethod
        def fetch_sequence(chrom, start, end):
            return g["chroms"][chrom][start:end]

    class R:
        chrom, start, end, summit = g["region"][0], g["region"][1], g["region"][2], (g["region"][1] + g["region"][2]) // 2

    s0 = scanner.Scanner(G, [R], window_size=0)
    assert [s0.sequences, s0.seq_starts, s0.seq_ends] == g["extract"]["w0"] and s0.window_size == 0
    s4 = scanner.Scanner(G, [R], window_size=4, strand="+")
    assert [s4.sequences, s4.seq_starts, s4.seq_ends] == g["extract"]["w4"]
    with pytest.raises(ValueError):
        scanner.Scanner(G, [R], window_size=0, strand="*")
    assert scanner.Scanner(G, [R], n_threads=0).n_threads == 1

    class P:
        matrix, cutoffs, length = np.array(g["pwm"], dtype=float), {"1e-3": 0.5}, 2

    with pytest.raises(ValueError):                           # missing cutoff: before any device work
        scanner.Scanner(G, [R], window_size=4, p_value="1e-2").scan_motifs([P])


def test_sweep_span_planning_is_pure_host_arithmetic():
    """ms_sweep_spans (no GPU needed): every window of every chromosome lies in exactly one span, spans respect the
    size bound, start on a window start, overlap their neighbour by window - stride, and number the windows globally."""
    rng = np.random.default_rng(5)
    for _ in range(60):
        window, stride = int(rng.integers(1, 300)), int(rng.integers(1, 120))
        lens = rng.integers(0, 20_000, size=int(rng.integers(1, 9))).tolist() + [window, max(window - 1, 0)]
        max_span = int(window + rng.integers(0, 6000))
        spans = _lib.sweep_spans(lens, window, stride, max_span)
        expect_first = 0
        by_chrom = {}
        for ch, b, e, first, n in spans:
            assert e - b <= max_span and b % stride == 0 and n >= 1 and (n - 1) * stride + window == e - b and e <= lens[ch]
            by_chrom.setdefault(ch, []).append((b, e, first, n))
            assert first == expect_first
            expect_first += n
        total = 0
        for ch, L in enumerate(lens):
            n_w = (L - window) // stride + 1 if L >= window else 0
            total += n_w
            got = by_chrom.get(ch, [])
            assert sum(x[3] for x in got) == n_w
            for (b0, e0, f0, n0), (b1, e1, f1, n1) in zip(got, got[1:]):
                assert b1 == b0 + n0 * stride and e0 - b1 == window - stride        # next span starts one stride after the last window
        assert expect_first == total
    with pytest.raises(ValueError):
        _lib.sweep_spans([100], 50, 10, 49)                     # a span must hold at least one window
    with pytest.raises(ValueError):
        _lib.sweep_spans([100], 0, 10, 100)


def test_streams_and_pinned_memory_fail_loudly_without_a_gpu():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    pw = _lib.PwmSet.from_matrices([np.zeros((4, 3))], cutoffs=[0.5])
    with pytest.raises(RuntimeError):
        _lib.Stream(pw)
    with pytest.raises(RuntimeError):
        _lib.PinnedBuffer(1024)


def test_measurement_switches_need_the_explicit_opt_in(monkeypatch):
    """MS_PF_* variables alone must not change what the library does (host-visible part: the plan)."""
    vals, widths, cutoffs = (np.load(os.path.join(ROOT, "tests", "golden", "synth_jaspar579.npz"))[k] for k in ("pwm_values", "widths", "cutoffs"))
    n = 40
    pw = _lib.PwmSet(vals[:4 * int(widths[:n].sum())], widths[:n], cutoffs[:n, 2])
    monkeypatch.delenv("MS_MEASURE", raising=False)
    monkeypatch.setenv("MS_PF_ENGINE", "0")
    base = pw.plan_mfma(3)                                       # engine 0 requested without the opt-in: still the matrix-core plan
    assert base["group_kb"].size > 0
    monkeypatch.setenv("MS_MEASURE", "1")
    with pytest.raises(ValueError):
        pw.plan_mfma(3)                                          # now engine 0 is really selected: no matrix-core plan to show
