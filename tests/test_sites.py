"""motifscan_amd.sites: the lazy nested view `Scanner.scan_motifs` returns behaves like the reference's nested lists
(/root/reference/motifscan/scanner.py:128-153) under everything its consumers do (io/__init__.py:23-54, stats.py:24-31)."""
import gc
import tracemalloc

import numpy as np
import pytest

from motifscan_amd.sites import LazyRegionSites, MotifSite, MotifSites, RegionRow, RegionSites


def _random_result(rng, P, R, n, max_pos=400):
    motif = np.sort(rng.integers(0, P, n))
    mo = np.concatenate([[0], np.cumsum(np.bincount(motif, minlength=P))]).astype(np.int64)
    region = np.empty(n, np.int64)
    for m in range(P):
        region[mo[m]:mo[m + 1]] = np.sort(rng.integers(0, R, mo[m + 1] - mo[m]))
    pos = rng.integers(0, max_pos, n).astype(np.int64)
    score = rng.random(n)
    strand = rng.integers(1, 3, n).astype(np.int8)
    starts = rng.integers(0, 10_000, R).astype(np.int64)
    eager = [[[] for _ in range(R)] for _ in range(P)]          # the reference's make_motif_sites, restated
    for k in range(n):
        eager[motif[k]][region[k]].append(MotifSite(int(starts[region[k]] + pos[k]), float(score[k]), "+" if strand[k] == 1 else "-"))
    return MotifSites(mo, region, pos, score, strand, starts), eager


@pytest.mark.parametrize("P,R,n", [(7, 50, 400), (3, 5, 0), (0, 4, 0), (2, 0, 0), (1, 1, 30), (5, 300, 9000)])
def test_lazy_view_equals_nested_lists(P, R, n):
    ms, eager = _random_result(np.random.default_rng(P * 1000 + R + n), P, R, n)
    assert len(ms) == P and all(len(per) == R for per in ms)
    assert ms == eager and eager == ms.to_lists() and not (ms != eager)
    assert [[len(x) for x in per] for per in ms] == [[len(x) for x in per] for per in eager]
    assert all(isinstance(per, RegionSites) for per in ms)
    for m in range(P):
        for r in range(R):
            got = ms[m][r]
            assert type(got) is tuple and list(got) == eager[m][r]
            assert all(type(s) is MotifSite and type(s.start) is int and type(s.score) is float and s.strand in "+-" for s in got)
    if P and R:
        assert list(ms[-1][-1]) == eager[-1][-1] and list(ms[0][np.int64(R - 1)]) == eager[0][R - 1]
        assert [list(x) for x in ms[0][1:4]] == eager[0][1:4] and [list(x) for x in ms[0][::-2]] == eager[0][::-2]
        assert [[list(x) for x in v] for v in ms[1:3]] == eager[1:3] and all(v == e for v, e in zip(ms[1:3], eager[1:3]))
        with pytest.raises(IndexError):
            ms[P]
        with pytest.raises(IndexError):
            ms[0][R]
        with pytest.raises(IndexError):
            ms[0][-R - 1]
    assert ms != eager + [[]] and ms != 3
    assert np.array_equal(ms.site_counts(), np.array([[len(x) for x in per] for per in eager], dtype=np.int32).reshape(P, R))
    want = np.array([[max(s.score for s in x) if x else np.nan for x in per] for per in eager], dtype=np.float64).reshape(P, R)
    assert np.array_equal(ms.max_scores(), want, equal_nan=True)
    assert np.array_equal(ms.n_regions_with_site, [sum(len(x) > 0 for x in per) for per in eager])
    for m in range(P):
        assert np.array_equal(ms[m].site_counts(), ms.site_counts()[m]) and ms[m].n_sites == sum(len(x) for x in eager[m])
        assert np.array_equal(ms[m].max_scores(), want[m], equal_nan=True)
    a = ms.arrays()
    flat = [(m, r, s.start, s.score, 1 if s.strand == "+" else 2) for m, per in enumerate(eager) for r, ss in enumerate(per) for s in ss]
    assert list(zip(a["motif"].tolist(), a["region"].tolist(), a["start"].tolist(), a["score"].tolist(), a["strand"].tolist())) == flat


def test_the_writers_and_the_statistics_read_it_like_lists():
    """io/__init__.py:23-33 (len(sites[idx]), max(site.score ...)), :47-54 (for site in sites[idx]) and stats.py:27-31, verbatim."""
    ms, eager = _random_result(np.random.default_rng(5), 6, 40, 500)

    def table(motif_sites, n_regions):
        rows = []
        for idx in range(n_regions):
            n_sites, scores = [], []
            for sites in motif_sites:
                num = len(sites[idx])
                n_sites.append(num)
                scores.append("NA" if num == 0 else max([site.score for site in sites[idx]]))
            rows.append((n_sites, scores))
        return rows

    def bed(motif_sites, n_regions):
        return [[(idx, site.start, site.score, site.strand) for idx in range(n_regions) for site in sites[idx]] for sites in motif_sites]

    def enrich(motif_sites):
        return [(len(sites), sum([len(s) > 0 for s in sites])) for sites in motif_sites]

    assert table(ms, 40) == table(eager, 40) and bed(ms, 40) == bed(eager, 40) and enrich(ms) == enrich(eager)
    for a, b in zip(ms, eager):                                  # zip(pwms, motif_sites, motif_sites_control): stats.py:24
        assert a == b
    # an item read twice in a row is one object (the writer reads len() then max()); region items are TUPLES: an edit fails loudly
    # instead of being lost (the view is read-only; to_lists() is the mutable form)
    r_first = eager[2].index(next(x for x in eager[2] if x))
    first = ms[2][r_first]
    assert first and ms[2][r_first] is first and type(first) is tuple
    with pytest.raises(AttributeError):
        first.append(None)
    with pytest.raises(AttributeError):
        ms[2][eager[2].index([])].append(None)
    # it pickles as what the reference returns: plain nested lists
    import pickle
    back = pickle.loads(pickle.dumps(ms))
    assert type(back) is list and back == eager and pickle.loads(pickle.dumps(ms[3])) == eager[3]


def test_nothing_of_size_pwms_x_regions_is_built_up_front():
    """SURVEY 8(b) 'lazily materialised for big R': 579 x 100 000 with 2M sites costs the view well under 1 MB and a millisecond
    scale construction; the reference's shape is 57.9M list objects (3.2 GB of empty lists)."""
    rng = np.random.default_rng(9)
    P, R, n = 579, 100_000, 2_000_000
    mo = np.linspace(0, n, P + 1).astype(np.int64)
    region = np.concatenate([np.sort(rng.integers(0, R, mo[m + 1] - mo[m])) for m in range(P)]).astype(np.int64)
    pos = rng.integers(0, 1000, n).astype(np.int64)
    score, strand, starts = rng.random(n), rng.integers(1, 3, n).astype(np.int8), np.arange(R, dtype=np.int64) * 1000
    gc.collect()
    tracemalloc.start()
    ms = MotifSites(mo, region, pos, score, strand, starts)
    assert len(ms) == P and len(ms[17]) == R
    built = tracemalloc.get_traced_memory()[0]
    assert built < 1 << 20, built
    want = [MotifSite(int(starts[region[k]] + pos[k]), float(score[k]), "+-"[strand[k] - 1]) for k in range(mo[17], mo[18]) if region[k] == 4242]
    assert list(ms[17][4242]) == want                                  # one motif's index: 4 bytes per region
    assert tracemalloc.get_traced_memory()[0] - built < 3 * R * 4 + (1 << 16)
    tracemalloc.stop()
    assert sum(len(x) > 0 for x in ms[3]) == len(np.unique(region[mo[3]:mo[4]]))


def test_iteration_hands_out_real_lists_and_indexing_stays_lazy():
    """`for sites in motif_sites` (io/__init__.py:26, stats.py:24) gets RegionRow objects -- real lists, so `sites[idx]` is the C list
    subscript -- built per motif on the first pass and kept; `motif_sites[m]` before any iteration is the lazy view; both compare
    equal to the reference's nested lists; when the rows would not fit the budget, iteration hands out the lazy views."""
    ms, eager = _random_result(np.random.default_rng(11), 9, 70, 900)
    assert type(ms[4]) is LazyRegionSites
    it = iter(ms)
    first = next(it)
    assert type(first) is RegionRow and isinstance(first, list) and isinstance(first, RegionSites) and type(ms[1]) is LazyRegionSites
    assert first == eager[0] and not (first != eager[0]) and first != eager[1]
    rest = list(it)
    assert all(type(r) is RegionRow for r in rest) and [first] + rest == eager
    assert all(a is b for a, b in zip(ms, [first] + rest))                  # the second pass: the same row objects, a plain list iterator
    assert type(iter(ms)).__name__ == "list_iterator"
    empty = next(x for x in first if not x)
    assert sum(x is empty for x in first) == sum(not x for x in eager[0])   # one shared empty tuple
    assert np.array_equal(first.site_counts(), [len(x) for x in eager[0]]) and first.n_sites == sum(len(x) for x in eager[0])
    assert ms == eager and ms.to_lists() == eager and type(ms.to_lists()[0][0]) is list
    small = MotifSites.row_budget_bytes
    try:
        MotifSites.row_budget_bytes = 100
        ms2, eager2 = _random_result(np.random.default_rng(11), 9, 70, 900)
        assert all(type(v) is LazyRegionSites for v in ms2) and ms2 == eager2
    finally:
        MotifSites.row_budget_bytes = small


def test_no_reference_cycle_and_close_releases_the_owner():
    """the views must not keep their container (and through it the scan result's pinned / device blocks) alive until
    the cycle collector runs.  Without gc: dropping the MotifSites drops the owner; close() does it while views are still held."""
    import weakref

    class Owner:
        pass

    rng = np.random.default_rng(3)
    ms, _ = _random_result(rng, 4, 20, 100)
    own = Owner()
    ref = weakref.ref(own)
    ms2 = MotifSites(ms._h.motif_offsets, ms._h.region, ms._h.pos, ms._h.score, ms._h.strand, ms._h.seq_starts, owner=own)
    del own
    gc.disable()
    try:
        rows = list(ms2)
        view = ms2[0]
        assert ref() is not None
        ms2.close()
        assert ref() is None and len(ms2) == 0                           # released with views and rows still referenced
        ms3 = MotifSites(ms._h.motif_offsets, ms._h.region, ms._h.pos, ms._h.score, ms._h.strand, ms._h.seq_starts, owner=Owner())
        r3 = weakref.ref(ms3._h.owner)
        list(ms3)
        del ms3
        assert r3() is None                                                # no cycle: plain reference counting frees it
        with MotifSites(ms._h.motif_offsets, ms._h.region, ms._h.pos, ms._h.score, ms._h.strand, ms._h.seq_starts) as m4:
            assert len(m4) == 4
        assert len(m4) == 0
    finally:
        gc.enable()
    del rows, view
