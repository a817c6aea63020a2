"""
CPU-only checks of the host side of libmotifscan_amd.so: the library loads and exports every
symbol the headers declare, argument validation maps to the reference's exception classes, the
pre-filter quantiser can never drop a window the reference reports, and the host-side
de-duplication equals the reference's.  No compute entry point is called (no GPU here).
"""
import ctypes
import os
import re
import sys

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

LUT = {c: i for i, c in enumerate("ACGT")}


def encode(seq):
    """2-bit codes and the non-ACGT mask of a sequence, as convert_seq sees it (cscore.c:92-111: case folded, the rest 'adds nothing')."""
    up = seq.upper()
    codes = np.array([LUT.get(c, 0) for c in up], dtype=np.int64)
    isn = np.array([c not in LUT for c in up], dtype=bool)
    return codes, isn


def field_strand(plan, n):
    return 1 + (n & 1) if plan["strand_mask"] == 3 else plan["strand_mask"]


def emulate_prefilter(plan, codes, isn):
    """numpy model of prefilter_f6_kernel, from the decoded PHYSICAL operand image: acc = bias + sum over the window's ACGT
    columns of rows[column][base] in units of 1/8 (a non-ACGT base is an all-zero one-hot column; the bias column is never
    cleared); a (field, window) is a candidate iff acc >= 0.  Returns {(motif, pos, strand)}."""
    L = len(codes)
    pc = np.concatenate([codes, np.zeros(70, dtype=np.int64)])
    pn = np.concatenate([isn, np.zeros(70, dtype=bool)])
    flagged = set()
    rows = plan["rows"].astype(np.int64)
    for q in range(rows.shape[0]):
        ncol = int(plan["group_cols"][q]) - 1                   # the last column of the group's fields carries the bias
        assert not rows[q, :, ncol:, :].any()
        for n in range(16):
            m = int(plan["group_fields"][q, n])
            acc = np.full(L, int(plan["bias"][q, n]), dtype=np.int64)
            for c in range(ncol):
                acc += np.where(pn[c:c + L], 0, rows[q, n, c][pc[c:c + L]])
            if plan["group_paired"][q]:                         # a paired row's field must stay inside its 11 bits (ms_internal.h)
                assert acc.min() >= -1024 and acc.max() < 1024
            hot = np.nonzero(acc >= 0)[0]
            if m < 0:
                assert len(hot) == 0                          # empty fields never flag
                continue
            sd = field_strand(plan, n)
            for j in hot:
                flagged.add((m, int(j), sd))
    return flagged


def check_plan_shape(plan, n_motifs, widths):
    assert plan["n_fast"] + plan["n_exact"] == n_motifs
    gf = plan["group_fields"]
    fast = set(gf.ravel().tolist()) - {-1}
    assert fast | set(plan["exact_motifs"].tolist()) == set(range(n_motifs)) and not (fast & set(plan["exact_motifs"].tolist()))
    kb, cols, paired = plan["group_kb"], plan["group_cols"], plan["group_paired"]
    q = 0
    while q < gf.shape[0]:                                      # a 32-row operand tile = two table groups, or four (fields X, Y) with paired rows
        n = 4 if paired[q] else 2
        assert q + n <= gf.shape[0] and (kb[q:q + n] == kb[q]).all() and (cols[q:q + n] == (8 if paired[q] else 16) * kb[q]).all()
        assert paired[q:q + n].tolist() == ([1, 2, 1, 2] if paired[q] else [0, 0])
        q += n
    for q in range(gf.shape[0]):
        for n in range(16):
            m = int(gf[q, n])
            if m >= 0:
                assert widths[m] <= cols[q] - 1                 # the motif's columns stay clear of the bias column
                assert widths[m] > 15 or paired[q]              # every motif of <= 15 columns rides a paired row ...
                assert widths[m] <= 23 or not paired[q]         # ... motifs of 16 ... 23 columns may (at 36 levels), wider ones never do
                if plan["strand_mask"] == 3:
                    assert gf[q, n ^ 1] == m                    # forward and reverse of a motif share a slot
    mag = np.abs(plan["rows"].astype(np.int64))                 # fp6 e2m3: every entry is on the grid (units of 1/8): sums are exact in f32
    assert ((mag <= 16) | ((mag <= 32) & (mag % 2 == 0)) | ((mag <= 60) & (mag % 4 == 0))).all()
    bm = np.abs(plan["bias"].astype(np.int64))
    assert ((bm <= 16) | ((bm <= 32) & (bm % 2 == 0)) | ((bm <= 60) & (bm % 4 == 0))).all()
    return fast


def random_seqs(rng, n, length, n_frac):
    """iid background sequences, some with runs of N (1 ... 40), some lower case."""
    out = []
    for i in range(n):
        s = rng.choice(list("ACGT"), p=[.295, .205, .205, .295], size=length)
        if n_frac and i % 2 == 0:
            for _ in range(max(1, int(n_frac * length / 12))):
                st = int(rng.integers(0, length - 1))
                s[st:st + int(rng.integers(1, 41))] = "N"
        s = "".join(s)
        out.append(s.lower() if i % 3 == 2 else s)
    return out


@pytest.mark.parametrize("pkey", ["1e-2", "1e-3", "1e-4"])
@pytest.mark.parametrize("strand", [1, 2, 3])
def test_prefilter_never_loses_a_reference_hit(oracle, rnd, pkey, strand):
    mats, cut = rnd["mats"], rnd["cutoff_by_key"][pkey]
    widths = [m.shape[1] for m in mats]
    pw = _lib.PwmSet.from_matrices(mats, cut)
    plan = pw.plan(strand)
    fast = check_plan_shape(plan, len(mats), widths)
    rng = np.random.default_rng(5)
    seqs = random_seqs(rng, 3, 3000, 0.0) + random_seqs(rng, 2, 2000, 0.03)       # N-free and with runs of N
    n_flag = n_hit = n_hit_n = 0
    for s in seqs:
        codes, isn = encode(s)
        flagged = emulate_prefilter(plan, codes, isn)
        assert all((strand >> (sd - 1)) & 1 for (_, _, sd) in flagged)      # a strand not asked for never flags
        sites = oracle.c_scan_motif([m.tolist() for m in mats], cut.tolist(), [s], strand, 2)
        for m, hits in enumerate(sites):
            if m not in fast:
                continue
            for _, pos, _, sd in hits:
                assert (m, pos, sd) in flagged, (m, pos, sd)
                n_hit += 1
                n_hit_n += bool(isn[pos:pos + widths[m]].any())
        if not isn.any():
            n_flag += sum(1 for (m, j, sd) in flagged if j + widths[m] <= len(s))
            n_hit_clean = n_hit
    assert n_hit_n > 0 or pkey == "1e-4"                        # windows with N do hit at the loose cutoffs
    # the filter must stay selective on N-free sequence (56 levels on the fp6 grid: coarse at p = 1e-2)
    assert n_flag <= 3.0 * n_hit_clean + 200, (n_flag, n_hit_clean)


def test_prefilter_keeps_hits_in_windows_with_non_acgt_bases(oracle, jaspar579):
    """N-aware operand: non-ACGT bases are all-zero one-hot columns, so windows with N go through the same filter.  Dense N
    (every window overlaps some) at the loosest cutoffs, where an all-N window can be a hit."""
    n = 60
    widths = jaspar579["widths"][:n]
    vals = jaspar579["pwm_values"][:4 * int(widths.sum())]
    mats, o = [], 0
    for w in widths:
        mats.append(vals[o:o + 4 * w].reshape(4, w))
        o += 4 * w
    rng = np.random.default_rng(11)
    for pkey, strand in (("1e-2", 3), ("1e-3", 1), ("1e-4", 2)):
        cut = jaspar579["cutoffs"][pkey][:n]
        pw = _lib.PwmSet(vals, widths, cut)
        plan = pw.plan(strand)
        fast = check_plan_shape(plan, n, widths)
        seqs = random_seqs(rng, 4, 1500, 0.08) + ["N" * 200, "ACGT" * 10 + "N" * 50 + "TTGACA" * 8, "NNNNNA" * 40]
        checked = 0
        for s in seqs:
            codes, isn = encode(s)
            flagged = emulate_prefilter(plan, codes, isn)
            sites = oracle.c_scan_motif([m.tolist() for m in mats], cut.tolist(), [s], strand, 2)
            for m, hits in enumerate(sites):
                if m in fast:
                    for _, pos, _, sd in hits:
                        assert (m, pos, sd) in flagged, (pkey, m, pos, sd)
                        checked += bool(isn[pos:pos + widths[m]].any())
            if set(s) == {"N"}:                                 # an all-N window flags only where the row bias alone is >= 0
                for q in range(plan["bias"].shape[0]):
                    for f in range(16):
                        m = int(plan["group_fields"][q, f])
                        if m >= 0 and plan["bias"][q, f] < 0:
                            assert not any((m, j, field_strand(plan, f)) in flagged for j in range(len(s) - 70))
        assert checked > 0 or pkey == "1e-4", pkey
    # at the CLI default the bias of most JASPAR-like motifs is negative: runs of N do not flood the candidate list
    pw = _lib.PwmSet(jaspar579["pwm_values"], jaspar579["widths"], jaspar579["cutoffs"]["1e-4"])
    plan = pw.plan(3)
    live = plan["group_fields"] >= 0
    assert (plan["bias"][live] < 0).mean() > 0.85


def test_plan_on_decision_boundary_cases(oracle):
    """The fuzzer's tie-heavy cases (cutoffs exactly on attainable scores, arbitrary matrices incl. columns whose best base
    is negative, sequences with non-ACGT bases): the plan keeps every hit."""
    import fuzz_parity
    checked = checked_n = 0
    for seed in range(60):
        mats, cutoffs, seqs, strand = fuzz_parity.make_case(seed)
        if len(mats) > 40:
            continue
        seqs = [s for s in seqs if len(s) >= 8][:6]
        if not seqs:
            continue
        widths = [m.shape[1] for m in mats]
        pw = _lib.PwmSet.from_matrices(mats, cutoffs)
        plan = pw.plan(strand)
        fast = check_plan_shape(plan, len(mats), widths)
        sites = oracle.c_scan_motif([m.tolist() for m in mats], cutoffs.tolist(), seqs, strand, 2)
        enc = [encode(s) for s in seqs]
        flagged = [emulate_prefilter(plan, c, n) for c, n in enc]
        for m, hits in enumerate(sites):
            if m in fast:
                for si, pos, _, sd in hits:
                    assert (m, pos, sd) in flagged[si], (seed, m, pos, sd)
                    checked += 1
                    checked_n += bool(enc[si][1][pos:pos + widths[m]].any())
    assert checked > 2000 and checked_n > 20, (checked, checked_n)


def test_prefilter_routes_degenerate_pwms_to_exact_path():
    wide = np.zeros((4, 70))
    wide[0] = 1.0
    allneg = -np.ones((4, 5))
    nonfinite = np.ones((4, 6))
    nonfinite[2, 3] = -np.inf
    ok = np.array([[1.0, -2, 0.5], [-1, 1.2, -0.3], [0.2, -0.4, 0.9], [-3, 0.1, -1.0]])
    low_cut = ok.copy()
    w40 = np.zeros((4, 40))
    w40[1] = 1.0
    pw = _lib.PwmSet.from_matrices([wide, allneg, nonfinite, ok, low_cut, w40], [0.5, 0.5, 0.5, 0.6, -50.0, 0.9])
    plan = pw.plan(3)
    assert sorted(plan["exact_motifs"].tolist()) == [0, 1, 2, 4]      # W > 63, max_raw = 0, -inf entry, every window passes
    assert plan["n_fast"] == 2                                     # a 3-column and a 40-column motif (3 k-blocks)
    kb_of = {int(plan["group_fields"][q, n]): int(plan["group_kb"][q]) for q in range(plan["group_fields"].shape[0]) for n in range(16)}
    assert kb_of[5] == 3 and kb_of[3] in (1, 3)


def test_plan_tiles_respect_lds_budget(jaspar579):
    """Motifs of <= 15 columns ride PAIRED rows: 32 motifs x {fwd, rev} (one strand: 64 motifs) per 32-row operand tile, four table
    groups, W // 8 + 1 half-blocks of 1.5 KiB; motifs of 16 ... 23 columns ride paired rows of three half-blocks (at 36 levels) as
    far as that lowers the instruction count; the rest plain rows: 16 (32) motifs per tile, two groups, W // 16 + 1 k-blocks;
    narrow to wide within each kind, the row tiles cut so that the instruction count is minimal; LDS tiles hold whole row tiles
    and stay inside the budget."""
    pw = _lib.PwmSet(jaspar579["pwm_values"], jaspar579["widths"], jaspar579["cutoffs"]["1e-4"])
    widths = np.asarray(jaspar579["widths"])
    n_pair, n_plain = int((widths <= 15).sum()), int((widths > 23).sum())       # at least / at least
    for strand, per_rt in ((3, 16), (1, 32), (2, 32)):
        for budget in (24 * 1024, 70 * 1024):
            plan = pw.plan(strand, budget)
            assert plan["n_exact"] == 0 and plan["n_fast"] == 579
            check_plan_shape(plan, 579, widths)
            gf, kb, tf, paired = plan["group_fields"], plan["group_kb"], plan["tile_first_group"], plan["group_paired"]
            rt_pair, rt_plain = int((paired == 1).sum()) // 2, int((paired == 0).sum()) // 2
            assert rt_pair >= (n_pair + 2 * per_rt - 1) // (2 * per_rt) and rt_plain >= (n_plain + per_rt - 1) // per_rt
            assert len(kb) == 4 * rt_pair + 2 * rt_plain
            assert (paired[:4 * rt_pair] > 0).all() and (paired[4 * rt_pair:] == 0).all()     # paired row tiles first
            assert (np.diff(kb[:4 * rt_pair]) >= 0).all() and (np.diff(kb[4 * rt_pair:]) >= 0).all() and kb.max() == 2 and kb.min() == 1
            assert sorted(set(gf[gf >= 0].tolist())) == list(range(579))
            assert (gf >= 0).sum() == 579 * (2 if strand == 3 else 1)
            assert plan["n_tiles"] == len(tf) - 1 and tf[-1] == len(kb) and (np.array(tf) % 2 == 0).all()
            first_of_rt = np.ones(len(kb), dtype=bool)             # one entry per row tile: its first group
            first_of_rt[:4 * rt_pair] = np.arange(4 * rt_pair) % 4 == 0
            first_of_rt[4 * rt_pair:] = np.arange(2 * rt_plain) % 2 == 0
            for t in range(len(tf) - 1):
                tile_bytes = int(kb[tf[t]:tf[t + 1]][first_of_rt[tf[t]:tf[t + 1]]].sum()) * 1536
                assert 0 < tile_bytes <= budget
            assert (plan["n_tiles"] == 1) == (budget > 64 * 1024)
    # the benchmark set at both strands: matrix instructions per 32 windows = the least over how many of the 16 ... 23-column motifs
    # (the narrowest first) ride paired rows, and over the cuts into runs of <= 32 / <= 16 motifs
    plan = pw.plan(3)

    def least(ws, per, cols):
        best = [0] + [10 ** 9] * len(ws)
        for i in range(1, len(ws) + 1):
            best[i] = min(best[j] for j in range(max(0, i - per), i)) + ws[i - 1] // cols + 1
        return best[-1]
    narrow, mid, rest = (sorted(int(w) for w in widths if lo <= w <= hi) for lo, hi in ((1, 15), (16, 23), (24, 63)))
    want = min(least(narrow + mid[:k], 32, 8) + least(sorted(mid[k:] + rest), 16, 16) for k in range(len(mid) + 1))
    paired = plan["group_paired"]
    rt_pair = int((paired == 1).sum()) // 2
    mid, rest = [], sorted(mid + rest)                      # (paired rows for 16 ... 23 columns: built, measured, switched off -- ms_internal.h)
    want = least(narrow, 32, 8) + least(rest, 16, 16)
    assert int(plan["group_kb"][:4 * rt_pair:4].sum()) + int(plan["group_kb"][4 * rt_pair::2].sum()) == want == 41


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


def test_motif_set_pipeline_in_one_pass_equals_the_reference_made_579(jaspar579):
    """The database-wide form (one [4, sum W] array, one numpy pass per stage) against the values the REFERENCE's
    per-motif objects produced: motifscan_amd/data/synth_jaspar579.npz was written by tests/golden/make_golden.py::make_579
    through PositionFrequencyMatrix.to_ppm().to_pwm(bg) (matrix.py:74-171) from this very seeded count stream."""
    rng = np.random.default_rng(20250310)
    widths = np.clip(np.rint(rng.gamma(shape=7.5, scale=1.55, size=579)), 5, 30).astype(np.int32)
    widths[0], widths[1] = 30, 5
    assert np.array_equal(widths, jaspar579["widths"])
    counts = []
    for w in widths:
        depth = rng.integers(20, 3001)
        c = np.rint(rng.dirichlet(0.3 * np.ones(4), size=int(w)).T * depth).astype(np.int64)
        c[:, c.sum(axis=0) == 0] = 1
        counts.append(c)
    pfms = matrix.MotifSet.from_matrices("pfm", counts, ids=[f"M{i}" for i in range(579)])
    pwms = pfms.to_ppm().to_pwm(dict(zip("ACGT", jaspar579["bg"])))
    vals, w = pwms.flat()
    assert np.array_equal(w, widths) and np.array_equal(vals, jaspar579["pwm_values"])          # bit for bit
    # a Motif is a window onto the set; the one-motif factories give the same numbers
    one = matrix.PositionFrequencyMatrix(counts[7]).to_ppm().to_pwm(dict(zip("ACGT", jaspar579["bg"])))
    assert np.array_equal(one.matrix, pwms[7].matrix) and pwms[7].length == widths[7] and pwms[-1].matrix_id == "M578"
    mx, mn = pwms.raw_extrema()
    assert float(mx[7]) == float(one.max_raw_score) == float(one.matrix.max(axis=0).sum()) and float(mn[7]) == float(one.min_raw_score)
    assert np.array_equal(pwms.max_raw_c(), _lib.PwmSet(vals, w, None).max_raw())               # cscore.c:36-48, the library's own
    with pytest.raises(ValueError):
        matrix.MotifSet.from_matrices("pfm", counts[:3] + [np.zeros((4, 2), dtype=np.int64)])


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


def test_views_of_library_memory_keep_their_owner_alive_through_numpy():
    """np.asarray() of a view (numpy collapses .base to the object that owns the memory), slices, reshapes and
    dtype views must all keep the owner -- the object whose release frees the memory -- alive; it goes when the last of them does."""
    import gc, weakref

    class Owner:
        def __init__(self, n):
            self.buf = (ctypes.c_int64 * n)(*range(n))

    o = Owner(16)
    ref = weakref.ref(o)
    v = _lib._owned_array(ctypes.addressof(o.buf), ctypes.c_int64, 16, np.int64, o)
    derived = [np.asarray(v), v[3:9], v.reshape(4, 4), v.view(np.uint8), np.asarray(v[2:])[::2]]
    del o, v
    gc.collect()
    assert ref() is not None
    assert derived[0][5] == 5 and derived[1][0] == 3 and derived[2][3, 3] == 15 and derived[4][1] == 4
    while derived:
        keep = derived.pop()
        gc.collect()
        assert ref() is not None or not derived
        del keep
    gc.collect()
    assert ref() is None


def test_bench_refuses_more_ranks_than_gpus_before_spawning():
    """`bench.py --gpus N` with N > visible devices exits non-zero with one line, before any rank is started or any data generated."""
    import subprocess, sys
    if _lib.device_count() >= 2:
        pytest.skip("two GPUs are visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MS_BENCH_SHARE_GPU")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0 and "nothing was started" in out.stderr and not out.stdout.strip()


def _check_isa():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "motifscan_amd", "csrc", "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_prefilter_isa_resources_and_the_atomic_register():
    """The build-time checks on the gfx950 code objects of the pre-filter are a MANDATORY step of
    the build since round 6 (csrc/check_isa.py, run by the Makefile right after each object is compiled: an object that fails is never
    linked; the rules are in that file's docstring).  Since round 6 the PRODUCT library holds no hand-written asm blocks at all (same
    speed, measured side by side: ms_kernels.hip "the two builds"); the blocks of rounds 4-5 live in the variant libmotifscan_amd_asm.so,
    and the rules they rest on -- operand registers that carry data in flight, the hand-out atomic's register, no scalar loads in the
    pass body -- guard THAT object.  This test re-runs both checks on the objects the two shipped libraries were linked from, and shows
    that the rules bite: the product object, which has none of the blocks, fails the asm rules."""
    ci = _check_isa()
    obj = os.path.join(ROOT, "motifscan_amd", "csrc", "ms_kernels.o")
    obj_asm = os.path.join(ROOT, "motifscan_amd", "csrc", "ms_kernels_asm.o")
    assert os.path.exists(obj), "the build leaves the object in csrc/ (it travels to the GPU box with the library)"
    if not ci.tools_present():
        pytest.skip("no ROCm llvm tools here: the build itself would have refused to link (check_isa.main)")
    assert _lib.lib().ms_build_flags() & 1 == (1 if _lib.LIB_VARIANT == "asm" else 0)
    print(ci.check(obj, no_asm=True))
    with pytest.raises(ci.IsaCheckError):
        ci.check(obj, no_asm=False)
    assert os.path.exists(obj_asm) == os.path.exists(os.path.join(ROOT, "motifscan_amd", "libmotifscan_amd_asm.so"))
    if os.path.exists(obj_asm):                                          # (absent only if this compiler tripped the asm rules: the build said so)
        print(ci.check(obj_asm, no_asm=False))


def test_both_library_variants_export_the_same_cabi():
    """libmotifscan_amd_asm.so (the variant with the hand-written blocks, csrc/Makefile) is the same library but for the pre-filter's object."""
    import shutil
    import subprocess
    nm = shutil.which("nm")
    if not nm:
        pytest.skip("no nm")
    if not os.path.exists(os.path.join(ROOT, "motifscan_amd", "libmotifscan_amd_asm.so")):
        pytest.skip("the asm variant was not built (its object failed the ISA check)")
    sets = []
    for name in ("libmotifscan_amd.so", "libmotifscan_amd_asm.so"):
        out = subprocess.run([nm, "-D", "--defined-only", os.path.join(ROOT, "motifscan_amd", name)], capture_output=True, text=True, check=True).stdout
        sets.append({l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("ms_")})
    assert sets[0] == sets[1] and "ms_build_flags" in sets[0]
    for variant, flag, so in (("", "0", "libmotifscan_amd.so"), ("asm", "1", "libmotifscan_amd_asm.so")):
        out = subprocess.run([sys.executable, "-c", "from motifscan_amd import _lib; print(_lib.lib().ms_build_flags(), _lib.LIB_PATH)"],
                             env=dict(os.environ, MS_LIB_VARIANT=variant, PYTHONPATH=ROOT), capture_output=True, text=True, check=True).stdout.split()
        assert out[0] == flag and out[1].endswith(so)


def test_measurement_switches_need_the_explicit_opt_in(monkeypatch):
    """measurement variables alone must not change what the library does; the retired engine / variant / tail switches
    of rounds 1-2 change nothing at all (host-visible part: the plan)."""
    vals, widths, cutoffs = (np.load(os.path.join(ROOT, "motifscan_amd", "data", "synth_jaspar579.npz"))[k] for k in ("pwm_values", "widths", "cutoffs"))
    n = 40
    pw = _lib.PwmSet(vals[:4 * int(widths[:n].sum())], widths[:n], cutoffs[:n, 2])
    monkeypatch.delenv("MS_MEASURE", raising=False)
    base = pw.plan(3)
    for var, val in (("MS_PF_ENGINE", "0"), ("MS_PF_VARIANT", "16"), ("MS_TAIL", "2"), ("MS_PF_FIELD_BITS", "16"), ("MS_PF_BQ_MAX", "100")):
        monkeypatch.setenv(var, val)
    for measure in (None, "1"):
        if measure:
            monkeypatch.setenv("MS_MEASURE", measure)
        other = _lib.PwmSet(vals[:4 * int(widths[:n].sum())], widths[:n], cutoffs[:n, 2]).plan(3)
        assert all(np.array_equal(base[k], other[k]) for k in ("group_fields", "rows", "bias", "group_kb"))


def test_integration_c_stub_of_the_collective_compiles(tmp_path):
    """how a compiled host runs the one collective on ms_result_region_counts_device's vector (RCCL ncclAllReduce,
    int64, 2 x n_pwms).  The block is extracted and compiled as plain C against include/ and the image's HIP / RCCL headers (no multi-GPU
    node here to run it on)."""
    import shutil, subprocess
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"A compiled host\*\* does \(b\).*?```c\n(.*?)```", text, re.S).group(1)
    if not (shutil.which("gcc") and os.path.exists("/opt/rocm/include/rccl/rccl.h")):
        pytest.skip("no gcc / RCCL headers")
    (tmp_path / "stub.c").write_text(code)
    out = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Wno-unused-result", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                          "-I" + os.path.join(ROOT, "include"), str(tmp_path / "stub.c")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "ncclAllReduce" in code and "ncclInt64" in code and "ms_result_region_counts_device" in code


def test_numa_lookups_against_a_fake_sysfs(tmp_path):
    """ms_numa.cpp (VERDICT r5 #3b): the NUMA node of a GPU comes from sysfs (numa_node of its PCI address), the node's CPUs from its
    cpulist; the stream's threads bind themselves there on multi-GPU nodes.  Here: the look-ups against a fabricated sysfs tree (two nodes,
    split CPU ranges, upper-case bus id, a device without NUMA information), and against this machine's own /sys (whatever it says, sane)."""
    L = _lib.lib()
    root = tmp_path
    dev = root / "sys/bus/pci/devices/0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    dev2 = root / "sys/bus/pci/devices/0000:05:00.0"
    dev2.mkdir(parents=True)
    (dev2 / "numa_node").write_text("-1\n")
    for n, cpus in ((0, "0-63,128-191"), (1, "64-127,192-255")):
        d = root / f"sys/devices/system/node/node{n}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    (root / "sys/devices/system/node/online").write_text("0-1\n")
    node, n_cpus, n_nodes = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    _lib.check(L.ms_debug_numa_probe(str(root).encode(), b"0000:C1:00.0", ctypes.byref(node), ctypes.byref(n_cpus), ctypes.byref(n_nodes)))
    assert (node.value, n_cpus.value, n_nodes.value) == (1, 128, 2)
    _lib.check(L.ms_debug_numa_probe(str(root).encode(), b"0000:05:00.0", ctypes.byref(node), ctypes.byref(n_cpus), ctypes.byref(n_nodes)))
    assert (node.value, n_cpus.value) == (-1, 0)
    _lib.check(L.ms_debug_numa_probe(str(root).encode(), b"0000:ff:00.0", ctypes.byref(node), ctypes.byref(n_cpus), ctypes.byref(n_nodes)))
    assert (node.value, n_cpus.value) == (-1, 0)
    _lib.check(L.ms_debug_numa_probe(b"", b"0000:00:00.0", ctypes.byref(node), ctypes.byref(n_cpus), ctypes.byref(n_nodes)))
    assert n_nodes.value >= 1 and node.value >= -1

