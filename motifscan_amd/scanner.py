"""
motifscan_amd.scanner -- host-side mirror of the reference's `motifscan.scanner`
(/root/reference/motifscan/scanner.py) over the MI355X scan path.

Same public surface and behaviour:
    MotifSite(start, score, strand)                      scanner.py:16
    Scanner(genome, regions, window_size=0, strand='both', p_value='1e-4',
            remove_dup=True, n_threads=1)                scanner.py:44-69
        .sequences / .seq_starts / .seq_ends / .window_size / .extend / .strand /
        .p_value / .remove_dup / .n_threads
        .scan_motifs(pwms) -> nested list [n_pwms][n_regions] of MotifSite   scanner.py:89-132
    make_motif_sites(sites, seq_starts)                  scanner.py:135-153
    deduplicate_motif_sites(motif_sites, lengths)        scanner.py:171-193
plus, for inputs where n_pwms x n_regions Python lists are not an option:
    Scanner.scan_motifs_arrays(pwms) -> flat numpy arrays in the same order.

`genome` is any object with `chrom_sizes[chrom]` and `fetch_sequence(chrom, start, end)`;
`regions` any objects with `.chrom .start .end .summit` -- i.e. the reference's own
`Genome` / `GenomicRegion` work unchanged.  `pwms` is any iterable of objects with
`.matrix` (4 x W), `.cutoffs[p_value]` and `.length`.
"""
import logging
import os

import numpy as np

from . import _lib
from .sites import MotifSite, MotifSites, RegionSites  # noqa: F401  (MotifSite is part of this module's surface, scanner.py:16)

logger = logging.getLogger(__name__)

_STRAND_FLAG = {"+": 1, "-": 2, "both": 3}


class _Frozen(list):
    """The scanner's region lists: real lists to read (==, len, index, slices), but the regions are FROZEN after construction --
    the packed device copy made on the first scan is kept for the scanner's life, so an edit would be silently ignored by
    scan_motifs while scan_batches re-encodes the strings.  Every mutating method raises instead."""
    __slots__ = ()

    def _refuse(self, *a, **k):
        raise TypeError("a Scanner's regions are frozen after construction (its packed device copy is cached): build a new Scanner")

    __setitem__ = __delitem__ = __iadd__ = __imul__ = append = extend = insert = pop = remove = clear = sort = reverse = _refuse


class Scanner:
    def __init__(self, genome, regions, window_size=0, strand="both", p_value="1e-4", remove_dup=True,
                 n_threads=1):
        self.window_size = window_size if window_size > 0 else 0
        self.extend = window_size // 2
        if strand not in _STRAND_FLAG:
            raise ValueError(f"invalid strand option: {strand!r}")
        self.strand = strand
        self.p_value = p_value
        self.remove_dup = remove_dup
        # kept for interface parity; the GPU path has no use for host threads (scanner.py:56-65)
        n_cpu = os.cpu_count() or 1
        n_threads = int(n_threads)
        if n_threads > n_cpu:
            logger.warning(f"Threads number exceed the number of CPUs, using {n_cpu} instead")
        self.n_threads = max(1, min(n_threads, n_cpu))
        self._starts, self._ends, self._sequences = [], [], []
        self._resident = None                # (ResidentGenome, chromosome indices) when extraction is on the device
        self._sq = None                      # the regions as a packed device set (0.375 B/base), kept between scan_motifs calls
        self._extract_seq(genome, regions)
        # frozen from here on: see _Frozen (the attributes are read-only properties, their lists refuse edits)
        self._starts, self._ends = _Frozen(self._starts), _Frozen(self._ends)
        if self._sequences is not None:
            self._sequences = _Frozen(self._sequences)

    @property
    def seq_starts(self):
        """0-based start of every scanned sequence on its chromosome (scanner.py:66)."""
        return self._starts

    @property
    def seq_ends(self):
        return self._ends

    @property
    def sequences(self):
        """Region sequences as strings (scanner.py:68).  With a ResidentGenome they are only
        materialised on demand (and only if the genome kept a host copy)."""
        if self._sequences is None:
            g, idx = self._resident
            self._sequences = _Frozen(g.fetch_sequence(g.names[c], lo, hi)
                                      for c, lo, hi in zip(idx, self.seq_starts, self.seq_ends))
        return self._sequences

    def _extract_seq(self, genome, regions):
        """Forward-strand sequence of every region (whole region, or a window of 2*(w//2) bp
        centred on the summit and clipped to the chromosome).  A `ResidentGenome` keeps the
        genome packed in HBM: then only coordinates are collected here and the regions are cut
        out on the device (ms_seqset_from_genome) instead of one fetch_sequence call per region."""
        logger.debug("Extracting sequences")
        whole = self.window_size <= 0
        resident = isinstance(genome, _lib.ResidentGenome)
        chrom_idx = []
        for region in regions:
            if whole:
                lo, hi = region.start, region.end
            else:
                lo = max(region.summit - self.extend, 0)
                hi = min(region.summit + self.extend, genome.chrom_sizes[region.chrom])
            self._starts.append(lo)
            self._ends.append(hi)
            if resident:
                chrom_idx.append(genome.index[region.chrom])
            else:
                self._sequences.append(genome.fetch_sequence(region.chrom, lo, hi))
        if resident:
            self._resident = (genome, chrom_idx)
            self._sequences = None

    def _seqset(self):
        """The regions as a device sequence set (convert_seq, cscore.c:81-114: 2-bit codes + non-ACGT mask), made on the first scan
        and kept for the scanner's life -- the reference converts its strings again on every c_scan_motif call.  The regions are
        frozen after construction (`_Frozen`), so the cached set cannot go stale; `close()` releases it."""
        if self._sq is None:
            if self._resident is not None:
                g, idx = self._resident
                self._sq = g.extract(idx, self.seq_starts, self.seq_ends)
            else:
                self._sq = _lib.SeqSet.from_strings(self._sequences)
        return self._sq

    def close(self):
        """Release the device copy of the regions (also done when the scanner is collected)."""
        if self._sq is not None:
            self._sq.close()
            self._sq = None

    def _as_sweep(self):
        """(genome, chromosome index, begin, end, window, stride) if the regions are the windows of ONE fixed-stride
        sweep of one chromosome of a ResidentGenome, in order -- then ms_scan_sweep gives the identical result while
        scoring every base once; None otherwise."""
        if self._resident is None or len(self.seq_starts) < 2:
            return None
        g, idx = self._resident
        st = np.asarray(self.seq_starts, dtype=np.int64)
        en = np.asarray(self.seq_ends, dtype=np.int64)
        ci = np.asarray(idx, dtype=np.int64)
        window, stride = int(en[0] - st[0]), int(st[1] - st[0])
        if window < 1 or stride < 1 or (ci != ci[0]).any() or ((en - st) != window).any() or (np.diff(st) != stride).any():
            return None
        return g, int(ci[0]), int(st[0]), int(en[-1]), window, stride

    def _as_overlapping(self):
        """(genome, chromosome indices) if the regions lie on a ResidentGenome and overlap enough (their union is under half of
        their summed length: peaks +- window/2 much closer than the window, dense random controls, cli/scan.py:43-48, 76-86) for
        ms_scan_regions_once to pay: it scores the union once and hands every site to each region that holds it -- the identical
        result.  At half it breaks even with the per-region scan (profiles/archive/r02_scan_once_overlap.log): the hand-out re-keys and
        re-orders every site."""
        if self._resident is None or len(self.seq_starts) < 2:
            return None
        g, idx = self._resident
        total = int(np.sum(np.asarray(self.seq_ends, dtype=np.int64) - np.asarray(self.seq_starts, dtype=np.int64)))
        if total == 0 or _lib.union_bases(idx, self.seq_starts, self.seq_ends) > 0.5 * total:
            return None
        return g, idx

    # ------------------------------------------------------------------ scanning --

    def _marshal(self, pwms):
        pwms = list(pwms)
        cutoffs = []
        for pwm in pwms:
            try:
                cutoffs.append(pwm.cutoffs[self.p_value])
            except (TypeError, KeyError):
                raise ValueError(f"PWM has no motif score cutoff set for P-value {self.p_value!r}")
        matrices = [np.asarray(pwm.matrix, dtype=np.float64) for pwm in pwms]
        lengths = [pwm.length for pwm in pwms]
        return matrices, np.asarray(cutoffs, dtype=np.float64), lengths

    def _scan(self, pwms, with_tables=False):
        """One device scan (+ de-dup): (hits dict of the library's pinned arrays, n_regions_with_site, tables or None)."""
        matrices, cutoffs, lengths = self._marshal(pwms)
        logger.debug("Scanning motif PWMs")
        pw = _lib.PwmSet.from_matrices(matrices, cutoffs)
        sweep = self._as_sweep()
        overlapping = self._as_overlapping() if sweep is None else None
        if sweep is not None:
            g, chrom, begin, end, window, stride = sweep
            res = _lib.scan_sweep(pw, g, chrom, begin, end, window, stride, _STRAND_FLAG[self.strand])
        elif overlapping is not None:
            g, idx = overlapping
            res = _lib.scan_regions_once(pw, g, idx, self.seq_starts, self.seq_ends, _STRAND_FLAG[self.strand])
        else:
            res = _lib.scan(pw, self._seqset(), _STRAND_FLAG[self.strand])
        try:
            if self.remove_dup:
                res.dedup(pw)                      # scanner.py:156-193 on the device, order preserved
            h = res.hits(copy=False, motif=False)  # views of the library's pinned host buffers (they keep `res` alive)
            region_counts = res.region_counts()
            tables = res.site_tables(len(self.seq_starts)) if with_tables else None
        except BaseException:
            res.close()
            raise
        finally:
            pw.close()
        return h, region_counts, tables, res

    def scan_motifs_arrays(self, pwms, with_tables=False):
        """Flat result: dict with motif, region, start (genome coordinate), score, strand (1/2),
        motif_offsets -- ordered exactly like the nested lists scan_motifs returns.  with_tables adds
        the dense per-(motif, region) site count and maximum score (NaN = no site)."""
        h, region_counts, tables, res = self._scan(pwms, with_tables)
        starts = np.asarray(self.seq_starts, dtype=np.int64)
        n_pwms = len(h["motif_offsets"]) - 1
        out = {"motif": np.repeat(np.arange(n_pwms, dtype=np.int32), np.diff(h["motif_offsets"])), "region": h["seq_idx"],
               "start": (starts[h["seq_idx"]] + h["pos"]) if len(h["pos"]) else h["pos"],
               "score": h["score"], "strand": h["strand"], "motif_offsets": h["motif_offsets"],
               "n_regions_with_site": region_counts,      # de-dup never empties a region
               "_result": res}                            # owns the memory behind region / score / strand
        if tables is not None:
            out["n_sites"], out["max_score"] = tables      # what write_sites_table prints (io/__init__.py:23-33)
        return out

    def scan_batches(self, pwms, batch_regions=100_000, depth=2):
        """Generator over consecutive batches of regions, for region lists whose sequences / hit lists should not sit in
        memory at once: yields (r0, r1, arrays) with `arrays` as scan_motifs_arrays returns them for regions [r0, r1)
        ('region' holds the GLOBAL region index).  The batches go through an ms_stream: the upload + packing of one batch,
        the scan of the previous one and the copy-out of the one before overlap on the device.  Concatenated per motif
        (_lib.merge_hits) the batches are exactly the single-call result."""
        matrices, cutoffs, lengths = self._marshal(pwms)
        n = len(self.seq_starts)
        pw = _lib.PwmSet.from_matrices(matrices, cutoffs)
        starts = np.asarray(self.seq_starts, dtype=np.int64)
        bounds = [(r0, min(n, r0 + batch_regions)) for r0 in range(0, n, max(1, int(batch_regions)))]

        def arrays(r0, res):
            h = res.hits(copy=False)
            region = h["seq_idx"] + r0
            return {"motif": h["motif"], "region": region, "start": (starts[region] + h["pos"]) if len(region) else h["pos"],
                    "score": h["score"], "strand": h["strand"], "motif_offsets": h["motif_offsets"],
                    "n_regions_with_site": res.region_counts(), "_result": res}     # score / strand are views of the result's pinned block

        def batches():
            for r0, r1 in bounds:
                seqs = [s.encode() if isinstance(s, str) else bytes(s) for s in self._sequences[r0:r1]]
                offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
                if seqs:
                    offsets[1:] = np.cumsum([len(b) for b in seqs])
                yield np.frombuffer(b"".join(seqs), dtype=np.uint8), offsets

        try:
            flags = _lib.MS_STREAM_DEDUP if self.remove_dup else 0
            if self._resident is not None:               # regions are cut on the device (ms_stream_submit_regions): nothing to upload
                g, idx = self._resident
                cuts = ((g, idx[r0:r1], self.seq_starts[r0:r1], self.seq_ends[r0:r1]) for r0, r1 in bounds)
                for (r0, r1), res in zip(bounds, _lib.scan_stream(pw, cuts, _STRAND_FLAG[self.strand], flags, depth)):
                    yield r0, r1, arrays(r0, res)
            else:
                for (r0, r1), res in zip(bounds, _lib.scan_stream(pw, batches(), _STRAND_FLAG[self.strand], flags, depth)):
                    yield r0, r1, arrays(r0, res)
        finally:
            pw.close()

    def count_regions_with_sites(self, pwms):
        """int64 [n_pwms]: the number of this scanner's regions that hold >= 1 site of each motif -- what
        `stats.motif_enrichment` computes from the nested lists (`sum(len(sites_by_region) > 0 ...)`, stats.py:29-31) and ALL the
        reference ever reads of the CONTROL regions' scan (cli/scan.py:81-89).  The hits stay on the device: no copy-out, no
        Python object per site.  (De-duplication never empties a region, so the count does not depend on remove_dup.)"""
        matrices, cutoffs, _ = self._marshal(pwms)
        pw = _lib.PwmSet.from_matrices(matrices, cutoffs)
        try:
            sweep = self._as_sweep()
            overlapping = self._as_overlapping() if sweep is None else None
            if sweep is not None:
                g, chrom, begin, end, window, stride = sweep
                res = _lib.scan_sweep(pw, g, chrom, begin, end, window, stride, _STRAND_FLAG[self.strand])
            elif overlapping is not None:
                g, idx = overlapping
                res = _lib.scan_regions_once(pw, g, idx, self.seq_starts, self.seq_ends, _STRAND_FLAG[self.strand])
            else:
                # nothing but the counts is read: the hits are counted where the fp64 stage left them -- no ordering, no site arrays
                res = _lib.scan(pw, self._seqset(), _STRAND_FLAG[self.strand], _lib.MS_SCAN_COUNTS_ONLY)
            try:
                return res.region_counts()
            finally:
                res.close()
        finally:
            pw.close()

    def scan_motifs(self, pwms):
        """motif_sites[n_pwms][n_regions] -> list[MotifSite] (scanner.py:89-132), as a read-only nested view over the
        device's flat hit arrays: a region's list is built when it is indexed (motifscan_amd/sites.py), so the call costs
        the scan + the copy-out whatever n_pwms x n_regions is.  `.to_lists()` gives the reference's real lists."""
        h, region_counts, _, res = self._scan(list(pwms))
        return MotifSites(h["motif_offsets"], h["seq_idx"], h["pos"], h["score"], h["strand"], self.seq_starts,
                          n_regions_with_site=region_counts, owner=res)


def make_motif_sites(sites, seq_starts):
    """Pooled [seq_idx, pos, score, strand] hits -> [n_pwms][n_seqs] lists of MotifSite in genome coordinates."""
    out = []
    for per_pwm in sites:
        buckets = [[] for _ in seq_starts]
        for seq_idx, pos, score, strand in per_pwm:
            buckets[seq_idx].append(MotifSite(seq_starts[seq_idx] + pos, score, "+" if strand == 1 else "-"))
        out.append(buckets)
    return out


def _walk_strand_in_given_order(sites, length):
    """One strand's sites of one region in the order the CALLER gave them: the survivor of each comparison meets the next
    site of the list; `next.start - current.start < length` also holds for a site that lies to the LEFT of the current one
    (scanner.py:156-168 does not sort).  Returns the kept sites in list order."""
    kept, cur = [], None
    for s in sites:
        if cur is None:
            cur = s
        elif s.start - cur.start < length:
            if not cur.score >= s.score:
                cur = s                                             # the later one scores strictly higher: it replaces the current site
        else:
            kept.append(cur)
            cur = s
    if cur is not None:
        kept.append(cur)
    return kept


def deduplicate_motif_sites(motif_sites, lengths):
    """Drop the lower-scoring one of two same-strand sites closer than the motif length (greedy walk over each strand's sites IN
    THE ORDER GIVEN, ties keep the earlier site); the result is ordered by start with '+' before '-' (scanner.py:156-193).
    Regions whose strands are start-sorted -- everything a scan produces -- go through the C routine the array path uses
    (ms_dedup_hits), all of them in one call; a region handed over in any other order is walked here, in its own order."""
    motif, region, start, score, strand = [], [], [], [], []
    out = [[[] for _ in per_pwm] for per_pwm in motif_sites]
    for m, (per_pwm, length) in enumerate(zip(motif_sites, lengths)):
        for r, sites in enumerate(per_pwm):
            fwd = [s for s in sites if s.strand == "+"]
            rev = [s for s in sites if s.strand != "+"]
            if all(a.start <= b.start for part in (fwd, rev) for a, b in zip(part, part[1:])):
                # (start asc, '+' first) inside a region is what the C routine wants; the stable merge keeps each strand's own order
                for s in sorted(fwd + rev, key=lambda s: s.start):
                    motif.append(m)
                    region.append(r)
                    start.append(s.start)
                    score.append(s.score)
                    strand.append(1 if s.strand == "+" else 2)
            else:
                kept = _walk_strand_in_given_order(fwd, length) + _walk_strand_in_given_order(rev, length)
                out[m][r] = sorted(kept, key=lambda s: s.start)
    n_pwms = len(motif_sites)
    offsets = np.concatenate([[0], np.cumsum(np.bincount(np.asarray(motif, dtype=np.int64), minlength=n_pwms))]) \
        if n_pwms else np.zeros(1)
    keep = _lib.dedup_keep(offsets, list(lengths), region, start, np.asarray(score, dtype=np.float64), strand) \
        if motif else np.zeros(0, dtype=bool)
    for k in np.nonzero(keep)[0].tolist():
        out[motif[k]][region[k]].append(MotifSite(start[k], score[k], "+" if strand[k] == 1 else "-"))
    return out
