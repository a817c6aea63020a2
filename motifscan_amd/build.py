"""
motifscan_amd.build -- the score-cutoff half of `motifscan motif --build` on the GPU
(/root/reference/motifscan/cli/motif.py:119-153 and motif/__init__.py:378-401).

The reference scores n_random background sequences with every PWM (`c_score`, both strands), sorts
each PWM's scores in descending order and takes, for e = 2 .. min(len(str(n)), 7) - 1, the score
at index int(n * 0.1**e) - 1 as the cutoff for P-value 1e-e; over `n_repeat` samplings the cutoffs
are averaged and rounded to 8 decimals.  Here the scoring, the sorting and the rank pick run on the
device (ms_score_ranks); only the P x (number of P-values) cutoffs come back.
"""
import numpy as np

from . import _lib


def cutoff_ranks(n_scores):
    """{'1e-2': rank, ...}: the 0-based ranks get_score_cutoffs reads for n_scores samples."""
    if n_scores < 100:
        raise ValueError("each motif must have at least 100 sampling scores")
    n_bits = min(len(str(n_scores)), 7)
    return {f"1e-{e}": int(n_scores * 0.1 ** e) - 1 for e in range(2, n_bits)}


def get_score_cutoffs(matrices, sequences, strand=3):
    """Per PWM a dict {p_value: cutoff} from ONE sampling (no averaging / rounding yet)."""
    pw = _lib.PwmSet.from_matrices(matrices)
    sq = _lib.SeqSet.from_strings(sequences)
    try:
        ranks = cutoff_ranks(sq.n_seqs)
        vals = _lib.score_ranks(pw, sq, list(ranks.values()), strand)
    finally:
        sq.close()
        pw.close()
    return [{k: float(vals[p, i]) for i, k in enumerate(ranks)} for p in range(len(matrices))]


def build_cutoffs(matrices, samplings, strand=3):
    """cli/motif.py:144-153: mean over the samplings' cutoffs, rounded to 8 decimals.
    samplings: list of sequence lists (one per repeat)."""
    per_repeat = [get_score_cutoffs(matrices, seqs, strand) for seqs in samplings]
    out = []
    for p in range(len(matrices)):
        keys = per_repeat[0][p].keys()
        out.append({k: float(np.around(np.mean([rep[p][k] for rep in per_repeat]), 8)) for k in keys})
    return out
