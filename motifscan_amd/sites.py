"""
motifscan_amd.sites -- the result of `Scanner.scan_motifs` as a read-only nested VIEW over flat hit arrays.

The reference returns `motif_sites[n_pwms][n_regions]`, a list of lists of lists of `MotifSite`
(/root/reference/motifscan/scanner.py:128-153): n_pwms x n_regions Python list objects before the first site is
looked at (4 GB / 30 s of empty lists at 579 x 100 000).  What its consumers do with it is
    len(motif_sites), zip / iterate over motifs                               stats.py:24-25, io/__init__.py:26,47
    len(sites), len(sites[idx]), max(site.score for site in sites[idx])       io/__init__.py:23-33, stats.py:27-31
    for site in sites[idx]: site.start / .score / .strand                     io/__init__.py:50-54
so `MotifSites` answers exactly those from the CSR arrays the device wrote (hits in the reference's order:
motif, region, position, '+' before '-'; `motif_offsets[P+1]`), and builds a region's `list[MotifSite]` only when
that region is indexed:

    MotifSites[m]            -> RegionSites (per-motif view; len == n_regions)
    MotifSites[m][r]         -> list[MotifSite]  (a plain list built on access, genome coordinates, '+' / '-'; asking for the SAME region
                                again right away returns the same list object -- `len(sites[idx])` then `max(... sites[idx])` --, any
                                other access builds a new one: the view is read-only, `to_lists()` is the mutable form)
    len(), iteration, negative indices, slices (-> plain lists of the items), == against plain nested lists

A motif's region index (`uint32[n_regions + 1]`, one bincount + cumsum over the motif's hits) is built the first time
one of ITS regions is indexed or iterated and kept: nothing of size n_pwms x n_regions exists unless every motif is
traversed, and then it is 4 bytes per (motif, region) against the reference's 56-byte empty list.  `to_lists()` gives
callers that want to mutate real lists the reference's eager form.  The vectorised accessors (`site_counts`,
`max_scores`, `n_regions_with_site`) serve the table writer and the enrichment statistics without any per-site object.
"""
from collections import namedtuple
from collections.abc import Sequence

import numpy as np

MotifSite = namedtuple("MotifSite", ["start", "score", "strand"])     # scanner.py:16

_STRAND_CHAR = (None, "+", "-")


def _index(i, n):
    if i < 0:
        i += n
    if not 0 <= i < n:
        raise IndexError("index out of range")
    return i


class RegionSites(Sequence):
    """motif_sites[m]: one motif's sites by region -- len == n_regions, item r == list[MotifSite]."""

    __slots__ = ("_p", "_m", "_lo", "_hi", "_off", "_off_np", "_last_r", "_last")

    def __init__(self, parent, m):
        self._p, self._m = parent, m
        mo = parent._motif_offsets
        self._lo, self._hi = int(mo[m]), int(mo[m + 1])
        self._off = self._off_np = self._last = None
        self._last_r = -1

    def _offsets(self):
        """uint32/uint64 [n_regions + 1] (as a memoryview: plain ints out): region r's sites are [lo + off[r], lo + off[r + 1])."""
        if self._off is None:
            n = self._p.n_regions
            off = np.zeros(n + 1, dtype=np.uint32 if self._hi - self._lo < 2 ** 32 else np.uint64)
            if self._hi > self._lo:
                np.cumsum(np.bincount(self._p._region[self._lo:self._hi], minlength=n), out=off[1:])
            self._off_np, self._off = off, memoryview(off)
        return self._off

    def _sites(self, a, b):
        p = self._p
        if b - a < 8:                                       # a region's few sites: plain ints out of memoryviews beat five numpy calls
            rg, ps, sc, sd, ss = p._mv
            return [MotifSite(ss[rg[k]] + ps[k], sc[k], _STRAND_CHAR[sd[k]]) for k in range(a, b)]
        reg, pos = p._region[a:b], p._pos[a:b]
        start = (p._seq_starts[reg] + pos).tolist()
        return [MotifSite(st, sc, _STRAND_CHAR[sd])
                for st, sc, sd in zip(start, p._score[a:b].tolist(), p._strand[a:b].tolist())]

    def __len__(self):
        return self._p.n_regions

    def __getitem__(self, r):
        off = self._off
        if off is None:
            off = self._offsets()
        if r.__class__ is not int:                          # slices, numpy integers: the slow lane
            if isinstance(r, slice):
                return [self[i] for i in range(*r.indices(self._p.n_regions))]
            r = r.__index__()
        if r < 0:
            r = _index(r, self._p.n_regions)
        a = off[r]
        b = off[r + 1]                                      # r >= n_regions: the memoryview raises IndexError
        if a == b:
            return []
        if r == self._last_r:                               # len(sites[idx]) then max(... sites[idx]) (io/__init__.py:28-33): one list
            return self._last
        self._last_r = r
        self._last = out = self._sites(self._lo + a, self._lo + b)
        return out

    def __iter__(self):
        """Every region's list in order: the motif's sites are made once, regions without a site cost one empty list."""
        self._offsets()
        counts = np.diff(self._off_np).tolist()
        sites = self._sites(self._lo, self._hi) if self._hi > self._lo else []
        k = 0
        for c in counts:
            if c:
                yield sites[k:k + c]
                k += c
            else:
                yield []

    def __eq__(self, other):
        if not isinstance(other, Sequence):
            return NotImplemented
        return len(other) == len(self) and all(a == b for a, b in zip(self, other))

    __hash__ = None

    def __repr__(self):
        return f"<RegionSites motif {self._m}: {self._hi - self._lo} sites in {self._p.n_regions} regions>"

    def __reduce__(self):
        return (list, (list(self),))

    # ---- without per-site objects ----
    @property
    def n_sites(self):
        return self._hi - self._lo

    def site_counts(self):
        """int64 [n_regions]: len(self[r]) for every r."""
        return np.bincount(self._p._region[self._lo:self._hi], minlength=self._p.n_regions)

    def max_scores(self):
        """float64 [n_regions]: max(site.score for site in self[r]), NaN where the region has no site."""
        out = np.full(self._p.n_regions, np.nan)
        if self._hi > self._lo:
            reg = self._p._region[self._lo:self._hi]
            first = np.flatnonzero(np.concatenate([[True], reg[1:] != reg[:-1]]))     # hits are ordered by region
            out[reg[first]] = np.maximum.reduceat(self._p._score[self._lo:self._hi], first)
        return out


class MotifSites(Sequence):
    """motif_sites: len == n_pwms, item m == RegionSites.  Built from the flat arrays of Scanner.scan_motifs_arrays."""

    def __init__(self, motif_offsets, region, pos, score, strand, seq_starts, n_regions_with_site=None, owner=None):
        self._motif_offsets = np.asarray(motif_offsets, dtype=np.int64)
        self._region, self._pos, self._score, self._strand = region, pos, score, strand
        self._seq_starts = np.asarray(seq_starts, dtype=np.int64)
        self._mv = tuple(memoryview(np.ascontiguousarray(x)) for x in (region, pos, score, strand, self._seq_starts))
        self.n_regions = len(self._seq_starts)
        self.n_pwms = len(self._motif_offsets) - 1
        self._n_regions_with_site = n_regions_with_site
        self._owner = owner                                 # whatever owns the memory behind the arrays
        # one view per motif, made once: the writers walk `for sites in motif_sites` once per REGION (io/__init__.py:26),
        # and a view keeps its motif's region index.  (Parent <-> view is a reference cycle: the cycle collector frees it.)
        self._views = [RegionSites(self, m) for m in range(self.n_pwms)]

    def __len__(self):
        return self.n_pwms

    def __getitem__(self, m):
        return self._views[m]                               # ints, negative ints, slices: a list's own rules

    def __iter__(self):
        return iter(self._views)

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
        return int(self._motif_offsets[-1])

    @property
    def n_regions_with_site(self):
        """int64 [n_pwms]: sum(len(s) > 0 for s in motif_sites[m]) -- what stats.py:29-31 computes."""
        if self._n_regions_with_site is None:
            self._n_regions_with_site = np.array([int((v.site_counts() > 0).sum()) for v in self], dtype=np.int64)
        return self._n_regions_with_site

    def arrays(self):
        """The flat form: motif, region, start (genome coordinate), score, strand (1 '+' / 2 '-'), motif_offsets."""
        return {"motif": np.repeat(np.arange(self.n_pwms, dtype=np.int32), np.diff(self._motif_offsets)),
                "region": self._region, "start": (self._seq_starts[self._region] + self._pos) if self.n_sites else self._pos,
                "score": self._score, "strand": self._strand, "motif_offsets": self._motif_offsets}

    def _cell(self):
        """int64 [n_sites]: motif * n_regions + region -- non-decreasing, because the hits are ordered by (motif, region)."""
        return np.repeat(np.arange(self.n_pwms, dtype=np.int64) * self.n_regions, np.diff(self._motif_offsets)) + self._region

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
            out[cell[first]] = np.maximum.reduceat(self._score, first)
        return out.reshape(self.n_pwms, self.n_regions)

    def to_lists(self):
        """The reference's eager form: real nested lists (n_pwms x n_regions list objects)."""
        return [list(v) for v in self]

    def __reduce__(self):
        """Pickles (multiprocessing, caches) as what the reference returns: plain nested lists -- the view itself sits on library-owned
        pinned memory that does not travel."""
        return (list, (self.to_lists(),))
