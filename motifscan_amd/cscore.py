"""
motifscan_amd.cscore -- drop-in for the reference's native module `motifscan.motif.cscore`
(/root/reference/motifscan/motif/cscore.c:479-494): the same two functions, the same argument
meaning and the same result shapes, computed on an MI355X through libmotifscan_amd.so.

    c_scan_motif(pwms, cutoffs, seqs, strand, n_threads) -> list[P] of list of [seq_idx, pos, score, strand]
    c_score(pwms, seqs, strand, n_threads)               -> list[P] of list[R] of float

`n_threads` is accepted for signature compatibility (cscore.c:404, 236) and ignored: the
parallel axis on the GPU is windows x motifs, not a pthread queue over PWMs.

Unlike the reference (which does no validation, cscore.c:117-120) malformed matrices raise
ValueError instead of crashing.
"""
import numpy as np

from . import _lib


def _check_strand(strand):
    strand = int(strand)
    if strand not in (1, 2, 3):
        raise ValueError(f"invalid strand flag: {strand!r} (1 forward, 2 reverse, 3 both)")
    return strand


def scan_arrays(pwm_values, widths, cutoffs, bases, offsets, strand=3, exact_only=False):
    """Flat-array form of c_scan_motif for large inputs.  Returns (hits dict, region_counts, stats)."""
    strand = _check_strand(strand)
    pw = _lib.PwmSet(pwm_values, widths, cutoffs)
    sq = _lib.SeqSet(bases, offsets)
    res = _lib.scan(pw, sq, strand, _lib.MS_SCAN_EXACT_ONLY if exact_only else _lib.MS_SCAN_DEFAULT)
    try:
        return res.hits(), res.region_counts(), res.stats()
    finally:
        res.close()
        sq.close()
        pw.close()


def c_scan_motif(pwms, cutoffs, seqs, strand, n_threads=1):
    strand = _check_strand(strand)
    if len(cutoffs) != len(pwms):
        raise ValueError("need one cutoff per PWM")
    pw = _lib.PwmSet.from_matrices(pwms, np.asarray(cutoffs, dtype=np.float64))
    sq = _lib.SeqSet.from_strings(seqs)
    res = _lib.scan(pw, sq, strand)
    try:
        h = res.hits()
    finally:
        res.close()
        sq.close()
        pw.close()
    off = h["motif_offsets"]
    seq, pos, sc, sd = h["seq_idx"].tolist(), h["pos"].tolist(), h["score"].tolist(), h["strand"].tolist()
    return [[[seq[k], pos[k], sc[k], sd[k]] for k in range(int(off[p]), int(off[p + 1]))]
            for p in range(len(pwms))]


def c_score(pwms, seqs, strand, n_threads=1):
    strand = _check_strand(strand)
    pw = _lib.PwmSet.from_matrices(pwms)
    sq = _lib.SeqSet.from_strings(seqs)
    try:
        return _lib.score(pw, sq, strand).tolist()
    finally:
        sq.close()
        pw.close()
