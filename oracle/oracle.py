"""
oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front end of the C restatement (oracle/cscore_oracle.c -> liboracle.so)
plus a plain-Python restatement of the reference's post-processing of the raw
hits (regrouping per region and the greedy overlap de-duplication).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; motifscan_amd/ never does.

Parity status: PINNED (see cscore_oracle.c header and tests/test_oracle_golden.py).

Reference lines restated here:
  c_scan_motif result shape   motifscan/motif/cscore.c:443-471
  c_score result shape        motifscan/motif/cscore.c:281-301
  make_motif_sites            motifscan/scanner.py:135-153
  _deduplicate_sites          motifscan/scanner.py:156-168
  deduplicate_motif_sites     motifscan/scanner.py:171-193
"""
import ctypes
import importlib.util
import os
import subprocess
import sys
from collections import namedtuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MotifSite = namedtuple("MotifSite", ["start", "score", "strand"])


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "cscore_oracle.c")
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src)
    if force or stale or not os.path.isdir(os.path.join(_HERE, "_ref")):
        subprocess.run(["make", "-s", "-C", _HERE, "all"], check=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if os.environ.get("ORACLE_SO"):                    # e.g. liboracle_asan.so (`make -C oracle asan`; tests/test_sanitizers.py)
            so = os.environ["ORACLE_SO"]
        elif not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        pd, pi32, pi64 = (ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32),
                          ctypes.POINTER(ctypes.c_int64))
        L.oracle_scan.restype = ctypes.c_int
        L.oracle_scan.argtypes = [pd, pi32, pd, ctypes.c_int32, ctypes.c_char_p, pi64, ctypes.c_int64,
                                  ctypes.c_int, ctypes.c_int, ctypes.POINTER(pi64), ctypes.POINTER(pi64),
                                  ctypes.POINTER(pi64), ctypes.POINTER(pd), ctypes.POINTER(pi32)]
        L.oracle_score.restype = ctypes.c_int
        L.oracle_score.argtypes = [pd, pi32, ctypes.c_int32, ctypes.c_char_p, pi64, ctypes.c_int64,
                                   ctypes.c_int, ctypes.c_int, pd]
        L.oracle_max_raw_score.restype = ctypes.c_double
        L.oracle_max_raw_score.argtypes = [pd, ctypes.c_int32]
        L.oracle_free.restype = None
        L.oracle_free.argtypes = [ctypes.c_void_p]
        L.oracle_encode.restype = None
        L.oracle_encode.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int8)]
        _LIB = L
    return _LIB


# ------------------------------------------------------------- marshalling --

def flatten_pwms(pwms):
    """list of 4xW array-likes -> (concatenated row-major values, widths)."""
    mats = [np.ascontiguousarray(np.asarray(m, dtype=np.float64)) for m in pwms]
    for m in mats:
        if m.ndim != 2 or m.shape[0] != 4:
            raise ValueError("each PWM must be 4 x W")
    widths = np.array([m.shape[1] for m in mats], dtype=np.int32)
    vals = (np.concatenate([m.ravel() for m in mats]) if mats else np.zeros(0)).astype(np.float64)
    return np.ascontiguousarray(vals), widths


def flatten_seqs(seqs):
    """list of str/bytes -> (concatenated bytes, int64 offsets[R+1])."""
    bs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in seqs]
    offsets = np.zeros(len(bs) + 1, dtype=np.int64)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs])
    return b"".join(bs), offsets


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def convert_seq(bases):
    """cscore.c:81-114 (convert_seq): bytes -> int8 codes, A/a 0, C/c 1, G/g 2, T/t 3, anything else -1 ("no contribution")."""
    raw = bytes(bases)
    out = np.zeros(len(raw), dtype=np.int8)
    lib().oracle_encode(raw, len(raw), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)))
    return out


def max_raw_score(matrix):
    m = np.ascontiguousarray(np.asarray(matrix, dtype=np.float64))
    return float(lib().oracle_max_raw_score(_p(m, ctypes.c_double), m.shape[1]))


def scan_arrays(vals, widths, cutoffs, bases, offsets, strand, n_threads=1):
    """Flat-array scan.  Returns dict of numpy arrays in the reference's hit order."""
    L = lib()
    P, R = len(widths), len(offsets) - 1
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    widths = np.ascontiguousarray(widths, dtype=np.int32)
    cutoffs = np.ascontiguousarray(cutoffs, dtype=np.float64)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    if isinstance(bases, np.ndarray):
        bases = bases.tobytes()
    po, ps, pp = (ctypes.POINTER(ctypes.c_int64)(), ctypes.POINTER(ctypes.c_int64)(),
                  ctypes.POINTER(ctypes.c_int64)())
    pv, pd = ctypes.POINTER(ctypes.c_double)(), ctypes.POINTER(ctypes.c_int32)()
    rc = L.oracle_scan(_p(vals, ctypes.c_double), _p(widths, ctypes.c_int32), _p(cutoffs, ctypes.c_double),
                       P, bases, _p(offsets, ctypes.c_int64), R, int(strand), int(n_threads),
                       ctypes.byref(po), ctypes.byref(ps), ctypes.byref(pp), ctypes.byref(pv),
                       ctypes.byref(pd))
    if rc != 0:
        raise MemoryError("oracle_scan failed")
    try:
        off = np.ctypeslib.as_array(po, shape=(P + 1,)).copy()
        n = int(off[-1])
        shape = (max(n, 1),)
        out = {
            "motif_offsets": off,
            "seq_idx": np.ctypeslib.as_array(ps, shape=shape)[:n].copy(),
            "pos": np.ctypeslib.as_array(pp, shape=shape)[:n].copy(),
            "score": np.ctypeslib.as_array(pv, shape=shape)[:n].copy(),
            "strand": np.ctypeslib.as_array(pd, shape=shape)[:n].copy(),
        }
    finally:
        for q in (po, ps, pp, pv, pd):
            L.oracle_free(ctypes.cast(q, ctypes.c_void_p))
    return out


def score_arrays(vals, widths, bases, offsets, strand, n_threads=1):
    L = lib()
    P, R = len(widths), len(offsets) - 1
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    widths = np.ascontiguousarray(widths, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    if isinstance(bases, np.ndarray):
        bases = bases.tobytes()
    out = np.zeros((P, R), dtype=np.float64)
    rc = L.oracle_score(_p(vals, ctypes.c_double), _p(widths, ctypes.c_int32), P, bases,
                        _p(offsets, ctypes.c_int64), R, int(strand), int(n_threads), _p(out, ctypes.c_double))
    if rc != 0:
        raise MemoryError("oracle_score failed")
    return out


# ----------------------------------------- reference-shaped entry points --

def c_scan_motif(pwms, cutoffs, seqs, strand, n_threads=1):
    """Same signature and result shape as the reference's cscore.c_scan_motif
    (cscore.c:399-476): list[P] of list of [seq_idx, pos, score, strand]."""
    vals, widths = flatten_pwms(pwms)
    bases, offsets = flatten_seqs(seqs)
    r = scan_arrays(vals, widths, np.asarray(cutoffs, dtype=np.float64), bases, offsets, strand, n_threads)
    off = r["motif_offsets"]
    out = []
    for p in range(len(widths)):
        a, b = int(off[p]), int(off[p + 1])
        out.append([[int(r["seq_idx"][k]), int(r["pos"][k]), float(r["score"][k]), int(r["strand"][k])]
                    for k in range(a, b)])
    return out


def c_score(pwms, seqs, strand, n_threads=1):
    """Same signature and result shape as the reference's cscore.c_score (cscore.c:231-302)."""
    vals, widths = flatten_pwms(pwms)
    bases, offsets = flatten_seqs(seqs)
    return score_arrays(vals, widths, bases, offsets, strand, n_threads).tolist()


# ------------------------------------------- scanner.py post-processing --

def make_motif_sites(sites, seq_starts):
    """scanner.py:135-153: bucket the pooled hits per region, shift to genome coordinates."""
    result = []
    for per_pwm in sites:
        buckets = [[] for _ in seq_starts]
        for seq_idx, pos, score, strand in per_pwm:
            buckets[seq_idx].append(MotifSite(seq_starts[seq_idx] + pos, score, "+" if strand == 1 else "-"))
        result.append(buckets)
    return result


def _dedup_one_strand(sites, length):
    """scanner.py:156-168: greedy left-to-right; on overlap (< length apart) drop the
    lower-scoring one, a tie keeps the earlier site; do not advance after a drop."""
    kept = list(sites)
    i = 0
    while i + 1 < len(kept):
        a, b = kept[i], kept[i + 1]
        if b.start - a.start < length:
            if a.score >= b.score:
                del kept[i + 1]
            else:
                del kept[i]
        else:
            i += 1
    return kept


def deduplicate_motif_sites(motif_sites, lengths):
    """scanner.py:171-193: per (motif, region), strands separately, then stable sort by start."""
    out = []
    for per_pwm, length in zip(motif_sites, lengths):
        regions = []
        for sites in per_pwm:
            fwd = _dedup_one_strand([s for s in sites if s.strand == "+"], length)
            rev = _dedup_one_strand([s for s in sites if s.strand != "+"], length)
            regions.append(sorted(fwd + rev, key=lambda s: s.start))
        out.append(regions)
    return out


# ------------------------------------------------- the real reference ext --

def load_reference_ext():
    """Import the REAL reference extension built into oracle/_ref (None if absent)."""
    d = os.path.join(_HERE, "_ref")
    if not os.path.isdir(d):
        return None
    for fn in sorted(os.listdir(d)):
        if fn.startswith("cscore") and fn.endswith(".so"):
            spec = importlib.util.spec_from_file_location("cscore", os.path.join(d, fn))
            try:
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                return mod
            except ImportError:
                return None
    return None


if __name__ == "__main__":
    build(force=True)
    print("oracle built;", "reference ext:", load_reference_ext() is not None, file=sys.stderr)
