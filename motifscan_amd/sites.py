"""
motifscan_amd.sites -- the result of `Scanner.scan_motifs` as a read-only nested VIEW over flat hit arrays.

The reference returns `motif_sites[n_pwms][n_regions]`, a list of lists of lists of `MotifSite`
(/root/reference/motifscan/scanner.py:128-153): n_pwms x n_regions Python list objects before the first site is
looked at (4 GB / 30 s of empty lists at 579 x 100 000).  What its consumers do with it is
    len(motif_sites), zip / iterate over motifs                               stats.py:24-25, io/__init__.py:26,47
    len(sites), len(sites[idx]), max(site.score for site in sites[idx])       io/__init__.py:23-33, stats.py:27-31
    for site in sites[idx]: site.start / .score / .strand                     io/__init__.py:50-54
so `MotifSites` answers exactly those from the CSR arrays the device wrote (hits in the reference's order:
motif, region, position, '+' before '-'; `motif_offsets[P+1]`).  Two lanes, one surface:

    MotifSites[m]            -> a per-motif sequence of length n_regions (`RegionSites`):
                                * INDEXED access builds nothing of size n_regions but the motif's 4-byte-per-region index, the
                                  first time one of its regions is asked for (`LazyRegionSites`);
                                * ITERATING over MotifSites -- what every consumer of the reference does (`for sites in
                                  motif_sites`, `zip(pwms, motif_sites, ...)`) -- hands out `RegionRow`s: real `list` objects, one
                                  slot per region, filled by ONE pass over the motif's hits (numpy boundaries + one zip), so that
                                  the writers' column-major walk `len(sites[idx])` costs a C list subscript, not a Python method
                                  call (round 4: 654 ns per (motif, region) cell through `__getitem__`).  Rows are built per motif
                                  when the iteration first reaches them and kept; when 8 bytes x n_pwms x n_regions plus the sites
                                  exceed `row_budget_bytes` (default 6 GiB) iteration falls back to the lazy views.
    MotifSites[m][r]         -> `tuple[MotifSite, ...]` in genome coordinates, '+' / '-' (`()` for a region without a site: ONE
                                shared object).  A tuple, not the reference's list: the view is read-only, and an `append` on it must
                                fail loudly instead of being lost; `to_lists()` is the mutable, eager form.
    len(), iteration, negative indices, slices (-> plain lists of the items), == against plain nested lists

Nothing here points back at its container: the views and rows hold the flat arrays (a small `_Hits` bundle), the `MotifSites`
holds the views -- no reference cycle, so dropping the result frees the pinned host / device blocks behind the arrays at once
(`close()` does it explicitly; `with scanner.scan_motifs(...) as sites:` works).  The vectorised accessors (`site_counts`,
`max_scores`, `n_regions_with_site`) serve the table writer and the enrichment statistics without any per-site object.
"""
import gc
from collections import namedtuple
from collections.abc import Sequence
from itertools import starmap

import numpy as np

MotifSite = namedtuple("MotifSite", ["start", "score", "strand"])     # scanner.py:16

_STRAND_CHAR = (None, "+", "-")
_NO_SITES = ()


def _index(i, n):
    if i < 0:
        i += n
    if not 0 <= i < n:
        raise IndexError("index out of range")
    return i


def _same_items(a, b):
    """Region items compare as sequences of sites, whatever holds them (our tuples, the reference's lists)."""
    return len(a) == len(b) and all(tuple(x) == tuple(y) for x, y in zip(a, b))


class _Hits:
    """The flat arrays of one result (and what keeps their memory alive).  Holds no view, no row, no MotifSites."""
    __slots__ = ("motif_offsets", "region", "pos", "score", "strand", "seq_starts", "mv", "n_regions", "owner")

    def __init__(self, motif_offsets, region, pos, score, strand, seq_starts, owner):
        self.motif_offsets = np.asarray(motif_offsets, dtype=np.int64)
        self.region, self.pos, self.score, self.strand = region, pos, score, strand
        self.seq_starts = np.asarray(seq_starts, dtype=np.int64)
        self.mv = tuple(memoryview(np.ascontiguousarray(x)) for x in (region, pos, score, strand, self.seq_starts))
        self.n_regions = len(self.seq_starts)
        self.owner = owner

    def sites(self, a, b):
        """hits [a, b) as MotifSite objects"""
        if b - a < 8:                                       # a region's few sites: plain ints out of memoryviews beat five numpy calls
            rg, ps, sc, sd, ss = self.mv
            return [MotifSite(ss[rg[k]] + ps[k], sc[k], _STRAND_CHAR[sd[k]]) for k in range(a, b)]
        start = (self.seq_starts[self.region[a:b]] + self.pos[a:b]).tolist()
        strand = np.array(_STRAND_CHAR, dtype=object)[self.strand[a:b]].tolist()
        return list(starmap(MotifSite, zip(start, self.score[a:b].tolist(), strand)))

    def drop(self):
        self.region = self.pos = self.score = self.strand = self.mv = self.owner = None


class RegionSites(Sequence):
    """motif_sites[m]: one motif's sites by region -- len == n_regions, item r == tuple of MotifSite.  The common surface of the
    lazy view and the materialised row (`RegionRow` is registered below)."""
    __slots__ = ()


class _MotifSlice:
    """What both forms know about their motif without per-site objects."""
    __slots__ = ()

    @property
    def n_sites(self):
        return self._hi - self._lo

    def site_counts(self):
        """int64 [n_regions]: len(self[r]) for every r."""
        return np.bincount(self._h.region[self._lo:self._hi], minlength=self._h.n_regions)

    def max_scores(self):
        """float64 [n_regions]: max(site.score for site in self[r]), NaN where the region has no site."""
        h = self._h
        out = np.full(h.n_regions, np.nan)
        if self._hi > self._lo:
            reg = h.region[self._lo:self._hi]
            first = np.flatnonzero(np.concatenate([[True], reg[1:] != reg[:-1]]))     # hits are ordered by region
            out[reg[first]] = np.maximum.reduceat(h.score[self._lo:self._hi], first)
        return out


class LazyRegionSites(_MotifSlice, RegionSites):
    """The indexed lane: region r's tuple is made when r is asked for; asking for the SAME region again right away returns the
    same object (`len(sites[idx])` then `max(... sites[idx])`, io/__init__.py:28-33)."""

    __slots__ = ("_h", "_m", "_lo", "_hi", "_off", "_off_np", "_last_r", "_last")

    def __init__(self, hits, m):
        self._h, self._m = hits, m
        self._lo, self._hi = int(hits.motif_offsets[m]), int(hits.motif_offsets[m + 1])
        self._off = self._off_np = self._last = None
        self._last_r = -1

    def _offsets(self):
        """uint32/uint64 [n_regions + 1] (as a memoryview: plain ints out): region r's sites are [lo + off[r], lo + off[r + 1])."""
        if self._off is None:
            n = self._h.n_regions
            off = np.zeros(n + 1, dtype=np.uint32 if self._hi - self._lo < 2 ** 32 else np.uint64)
            if self._hi > self._lo:
                np.cumsum(np.bincount(self._h.region[self._lo:self._hi], minlength=n), out=off[1:])
            self._off_np, self._off = off, memoryview(off)
        return self._off

    def __len__(self):
        return self._h.n_regions

    def __getitem__(self, r):
        off = self._off
        if off is None:
            off = self._offsets()
        if r.__class__ is not int:                          # slices, numpy integers: the slow lane
            if isinstance(r, slice):
                return [self[i] for i in range(*r.indices(self._h.n_regions))]
            r = r.__index__()
        if r < 0:
            r = _index(r, self._h.n_regions)
        a = off[r]
        b = off[r + 1]                                      # r >= n_regions: the memoryview raises IndexError
        if a == b:
            return _NO_SITES
        if r == self._last_r:
            return self._last
        self._last_r = r
        self._last = out = tuple(self._h.sites(self._lo + a, self._lo + b))
        return out

    def __iter__(self):
        return iter(build_row(self._h, self._lo, self._hi))

    def __eq__(self, other):
        if not isinstance(other, Sequence):
            return NotImplemented
        return _same_items(self, other)

    __hash__ = None

    def __repr__(self):
        return f"<RegionSites motif {self._m}: {self._hi - self._lo} sites in {self._h.n_regions} regions>"

    def __reduce__(self):
        return (list, ([list(x) for x in self],))


class RegionRow(_MotifSlice, list):
    """The iterated lane: a real list, slot r == region r's tuple of MotifSite (`()` shared by every empty region), so that
    subscripting, len() and iteration are the C list's own.  Knows its motif's slice of the flat arrays for the vectorised
    accessors; holds no reference to the MotifSites it came from."""
    __slots__ = ("_h", "_lo", "_hi")

    def __eq__(self, other):
        if not isinstance(other, Sequence):
            return NotImplemented
        return _same_items(self, other)

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None

    def __reduce__(self):
        return (list, ([list(x) for x in self],))


RegionSites.register(RegionRow)


def build_row(hits, lo, hi):
    """One motif's RegionRow: n_regions slots in one C-level fill, then the motif's sites made ONCE (one zip over the slice) and
    cut at the region boundaries numpy finds -- regions without a site cost nothing beyond their slot."""
    row = RegionRow([_NO_SITES] * hits.n_regions)
    row._h, row._lo, row._hi = hits, lo, hi
    if hi > lo:
        reg = hits.region[lo:hi]
        first = np.flatnonzero(np.concatenate([[True], reg[1:] != reg[:-1]]))
        bounds = np.concatenate([first, [hi - lo]]).tolist()
        # millions of small tuples: with the cycle collector on, every generation-2 pass walks all of them again (measured: 42 s
        # instead of 6 s for 9M sites); none of these objects can be part of a cycle
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            sites = hits.sites(lo, hi)
            for r, a, b in zip(reg[first].tolist(), bounds, bounds[1:]):
                row[r] = tuple(sites[a:b])
        finally:
            if gc_was_on:
                gc.enable()
    return row


class MotifSites(Sequence):
    """motif_sites: len == n_pwms, item m == RegionSites.  Built from the flat arrays of Scanner.scan_motifs_arrays."""

    row_budget_bytes = 6 << 30        # iteration materialises rows while 8 B x n_pwms x n_regions + ~160 B per site stays below this

    def __init__(self, motif_offsets, region, pos, score, strand, seq_starts, n_regions_with_site=None, owner=None):
        self._h = h = _Hits(motif_offsets, region, pos, score, strand, seq_starts, owner)
        self.n_regions = h.n_regions
        self.n_pwms = len(h.motif_offsets) - 1
        self._n_regions_with_site = n_regions_with_site
        # one view per motif, made once: a view keeps its motif's region index.  The views know the arrays, not this object.
        self._views = [LazyRegionSites(h, m) for m in range(self.n_pwms)]
        self._rows_left = self.n_pwms           # rows still to be built before iteration is the plain list iterator
        self._use_rows = 8 * self.n_pwms * self.n_regions + 160 * int(h.motif_offsets[-1]) <= self.row_budget_bytes

    # ---- lifetime ----
    def close(self):
        """Drop the arrays (and with them the pinned host / device blocks of the scan result) now; the object is empty afterwards."""
        self._h.drop()
        self._views = []
        self.n_pwms = self._rows_left = 0

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- the reference's surface ----
    def __len__(self):
        return self.n_pwms

    def __getitem__(self, m):
        return self._views[m]                               # ints, negative ints, slices: a list's own rules

    def _row(self, m):
        v = self._views[m]
        if v.__class__ is not RegionRow:
            self._views[m] = v = build_row(self._h, v._lo, v._hi)
            self._rows_left -= 1
        return v

    def _first_pass(self):
        """Builds each motif's row as the iteration gets there.  The cycle collector stays ON for the consumer's loop bodies (ADVICE r5:
        it used to be off for the whole pass, i.e. for whatever the consumer did between two rows, for as long as the iterator lived).
        What made the pass quadratic -- every full collection walking the n_regions slots and the sites of every row built so far (measured,
        579 motifs x 100 000 regions x 9M sites: 18 s against 6 s) -- is handled where it arises: build_row makes a row with the collector
        off, and the finished row is moved to the collector's permanent generation (gc.freeze(): nothing a row holds can be part of a
        cycle, reference counting frees it).  freeze() takes whatever else is alive along; the finally clause -- which also runs when the
        consumer abandons the pass (generator close) -- hands everything back to the ordinary generations."""
        try:
            for m in range(self.n_pwms):
                row = self._row(m)
                gc.freeze()
                yield row
        finally:
            gc.unfreeze()

    def __iter__(self):
        if self._use_rows and self._rows_left:
            return self._first_pass()
        return iter(self._views)                            # rows all built: the C list iterator over real lists (or: lazy views)

    def __eq__(self, other):
        if not isinstance(other, Sequence):
            return NotImplemented
        return len(other) == len(self) and all(a == b for a, b in zip(self, other))

    __hash__ = None

    def __repr__(self):
        return f"<MotifSites {self.n_pwms} motifs x {self.n_regions} regions, {self.n_sites} sites>"

    # ---- flat / vectorised access ----
    @property
    def n_sites(self):
        return int(self._h.motif_offsets[-1])

    @property
    def n_regions_with_site(self):
        """int64 [n_pwms]: sum(len(s) > 0 for s in motif_sites[m]) -- what stats.py:29-31 computes."""
        if self._n_regions_with_site is None:
            self._n_regions_with_site = np.array([int((v.site_counts() > 0).sum()) for v in self._views], dtype=np.int64)
        return self._n_regions_with_site

    def arrays(self):
        """The flat form: motif, region, start (genome coordinate), score, strand (1 '+' / 2 '-'), motif_offsets."""
        h = self._h
        return {"motif": np.repeat(np.arange(self.n_pwms, dtype=np.int32), np.diff(h.motif_offsets)),
                "region": h.region, "start": (h.seq_starts[h.region] + h.pos) if self.n_sites else h.pos,
                "score": h.score, "strand": h.strand, "motif_offsets": h.motif_offsets}

    def _cell(self):
        """int64 [n_sites]: motif * n_regions + region -- non-decreasing, because the hits are ordered by (motif, region)."""
        return np.repeat(np.arange(self.n_pwms, dtype=np.int64) * self.n_regions, np.diff(self._h.motif_offsets)) + self._h.region

    def site_counts(self):
        """int32 [n_pwms][n_regions]: len(motif_sites[m][r]) (io/__init__.py:28-29)."""
        n = self.n_pwms * self.n_regions
        return np.bincount(self._cell(), minlength=n).astype(np.int32).reshape(self.n_pwms, self.n_regions)

    def max_scores(self):
        """float64 [n_pwms][n_regions]: max(site.score ...), NaN = 'NA' (io/__init__.py:30-33)."""
        out = np.full(self.n_pwms * self.n_regions, np.nan)
        if self.n_sites:
            cell = self._cell()
            first = np.flatnonzero(np.concatenate([[True], cell[1:] != cell[:-1]]))
            out[cell[first]] = np.maximum.reduceat(self._h.score, first)
        return out.reshape(self.n_pwms, self.n_regions)

    def to_lists(self):
        """The reference's eager, mutable form: real nested lists of lists (n_pwms x n_regions list objects)."""
        return [[list(x) for x in build_row(self._h, v._lo, v._hi)] for v in self._views]

    def __reduce__(self):
        """Pickles (multiprocessing, caches) as what the reference returns: plain nested lists -- the view itself sits on library-owned
        pinned memory that does not travel."""
        return (list, (self.to_lists(),))
