"""
motifscan_amd.matrix -- motif matrices as ONE struct-of-arrays set, built for the device.

The reference keeps one Python object per motif and per stage (counts -> probabilities -> log-odds,
/root/reference/motifscan/motif/matrix.py:33-240) and walks them one by one.  The scan kernel wants the opposite: every
motif of a database as one flat buffer (`ms_pwmset_create`: values [sum 4*W], widths [P]).  So the unit here is the
`MotifSet`: all columns of all motifs side by side in one `[4, sum W]` array with a `widths` vector, a `kind` tag
('pfm' counts / 'ppm' probabilities / 'pwm' log-odds), and the three stage transforms as ONE numpy pass over every
column of the database (579 JASPAR motifs = 6.7k columns = one ufunc call each).  What the numbers mean is the
reference's definition, pinned by the reference-made G7 golden (tests/test_host_cabi.py):

    columns_to_ppm    counts / column sum; columns holding a 0 get pseudo/(1-4*pseudo) added, then renormalised
                      (matrix.py:74-98, 125-147;  [0,0,10,10] -> [0.001,0.001,0.499,0.499])
    columns_to_pwm    around(ln(ppm / bg), 5)                                          (matrix.py:149-171)
    MotifSet.raw_extrema   numpy column extrema, NOT clamped at 0                       (matrix.py:202-214)
    MotifSet.max_raw_c     the C scorer's clamped form, what the kernel divides by      (cscore.c:36-48)
    MotifSet.score / Motif.score   scalar forward-strand score of one W-mer            (matrix.py:216-240)
    MotifSet.score_batch   (new) every motif x many sequences on the GPU (`ms_score`, the c_score kernel)

`Motif` is a one-motif window onto a set with the attributes the rest of the path reads (`.matrix`, `.length`,
`.cutoffs`, `.name`, `.matrix_id`: scanner.py:110-123, io/__init__.py:23-33); the reference's three constructor names
are kept as factory functions at the bottom so that its call sites read the same.  Error messages follow the
reference's wording (they are part of its tested behaviour, tests/test_motif_matrix.py:9-60).
"""
import numpy as np

BASES = "ACGT"
KINDS = ("any", "pfm", "ppm", "pwm")
_CODE = np.full(256, -1, dtype=np.int8)
for _i, _b in enumerate(BASES):
    _CODE[ord(_b)] = _CODE[ord(_b.lower())] = _i


# ------------------------------------------------------------------ column-level rules (vectorised over a database) --

def _as_columns(values):
    """One motif's 4 x W input -> a 2-d numeric ndarray, or the reference's ValueError."""
    if len(values) != 4:
        raise ValueError("values should have exactly 4 rows for A/C/G/T")
    cols = np.asarray(values)
    if cols.ndim != 2:
        raise ValueError("values should have 2 dimensions in (4 x N)")
    if cols.dtype.kind not in "iuf":
        raise ValueError("values should be integers or floating numbers")
    if cols.shape[1] < 1:
        raise ValueError("values should have at least 1 position per row")
    return cols


def check_columns(kind, cols):
    """The per-stage admissibility rules, on any number of columns at once (matrix.py:45-52, 106-116)."""
    if kind == "pfm":
        if cols.dtype.kind not in "iu" or cols.min() < 0:
            raise ValueError("values in PFM should be non-negative integers")
        if not cols.sum(axis=0).all():
            raise ValueError("all values of a PFM position are 0")
    elif kind == "ppm":
        if cols.min() < 0:
            raise ValueError("values in PPM should be non-negative numbers")
        total = cols.sum(axis=0)
        if not total.all():
            raise ValueError("all values of a PPM position are 0")
        if not np.allclose(total, 1):
            raise ValueError("the sum probability of a PPM position is not 1")


def pseudo_normalize(ppm_cols, pseudo=0.001):
    """Columns with an impossible base are lifted by pseudo/(1-4*pseudo) and every column rescaled to sum 1."""
    if not 0 < pseudo < 0.25:
        raise ValueError("the range of pseudo should be (0, 0.25)")
    lift = np.where((ppm_cols == 0).any(axis=0), pseudo / (1 - 4 * pseudo), 0.0)
    lifted = ppm_cols + lift
    return lifted / lifted.sum(axis=0)


def columns_to_ppm(count_cols, normalize=True, pseudo=0.001):
    ppm = count_cols / count_cols.sum(axis=0)
    return pseudo_normalize(ppm, pseudo) if normalize else ppm


def columns_to_pwm(ppm_cols, bg_freq=None):
    bg = np.array([0.25] * 4 if bg_freq is None else [bg_freq[b] for b in BASES], dtype=np.float64)
    return np.around(np.log(ppm_cols / bg[:, None]), 5)


# ----------------------------------------------------------------------------------------------------- the set --

class MotifSet:
    """P motifs as one `[4, sum W]` array (`cols`), `widths[P]`, `starts[P+1]`; `kind` names the stage."""

    def __init__(self, kind, cols, widths, names=None, ids=None, cutoffs=None, _checked=False):
        assert kind in KINDS
        self.kind = kind
        self.cols = cols
        self.widths = np.asarray(widths, dtype=np.int32)
        self.starts = np.concatenate([[0], np.cumsum(self.widths, dtype=np.int64)])
        n = len(self.widths)
        self.names = list(names) if names is not None else [None] * n
        self.ids = list(ids) if ids is not None else [None] * n
        self.cutoffs = list(cutoffs) if cutoffs is not None else [None] * n       # per motif: dict p_value -> cutoff, or None
        if not _checked and n:
            check_columns(kind, cols)

    @classmethod
    def from_matrices(cls, kind, matrices, names=None, ids=None, cutoffs=None):
        mats = [_as_columns(m) for m in matrices]
        cols = np.concatenate(mats, axis=1) if mats else np.zeros((4, 0))
        return cls(kind, cols, [m.shape[1] for m in mats], names, ids, cutoffs)

    def __len__(self):
        return len(self.widths)

    def __getitem__(self, i):
        return Motif(self, range(len(self))[i])

    def __iter__(self):
        return (Motif(self, i) for i in range(len(self)))

    def _derive(self, kind, cols):
        return MotifSet(kind, cols, self.widths, self.names, self.ids, [None if c is None else dict(c) for c in self.cutoffs], _checked=True)

    def to_ppm(self, normalize=True, pseudo=0.001):
        return self._derive("ppm", columns_to_ppm(self.cols, normalize, pseudo))

    def normalized(self, pseudo=0.001):
        return self._derive("ppm", pseudo_normalize(self.cols, pseudo))

    def to_pwm(self, bg_freq=None):
        return self._derive("pwm", columns_to_pwm(self.cols, bg_freq))

    def _per_motif_sum(self, per_col):
        # ndarray.sum() per motif: numpy's pairwise order, so the value is bit-for-bit the reference property's
        return np.array([per_col[a:b].sum() for a, b in zip(self.starts[:-1], self.starts[1:])], dtype=np.float64)

    def raw_extrema(self):
        """(max_raw[P], min_raw[P]): sums of the column maxima / minima, numpy's definition (matrix.py:202-214)."""
        return self._per_motif_sum(self.cols.max(axis=0)), self._per_motif_sum(self.cols.min(axis=0))

    def max_raw_c(self):
        """The C scorer's max_raw: every column maximum clamped at 0 and summed left to right (cscore.c:36-48) -- the
        library computes its own (ms_pwmset_create); this is the host's copy for planning and tests."""
        best = np.maximum(self.cols.max(axis=0), 0.0)
        return np.array([sum(best[a:b].tolist(), 0.0) for a, b in zip(self.starts[:-1], self.starts[1:])])

    def flat(self):
        """(values [sum 4*W] motif after motif, row-major 4 x W each; widths): `ms_pwmset_create`'s operands."""
        vals = np.concatenate([self.cols[:, a:b].ravel() for a, b in zip(self.starts[:-1], self.starts[1:])]) if len(self) else np.zeros(0)
        return np.ascontiguousarray(vals, dtype=np.float64), self.widths

    def cutoff_vector(self, p_value):
        return np.array([c[p_value] for c in self.cutoffs], dtype=np.float64)

    def score(self, i, sequence):
        """raw / max_raw of one sequence of exactly motif i's width, forward strand; other letters add nothing."""
        a, w = int(self.starts[i]), int(self.widths[i])
        if len(sequence) != w:
            raise ValueError("sequence should have the same length as the PWM")
        block = self.cols[:, a:a + w]
        code = _CODE[np.frombuffer(sequence.encode("latin-1", "replace"), dtype=np.uint8)]
        known = np.nonzero(code >= 0)[0]
        raw = 0
        for c in known:                                       # left to right, like the scalar reference
            raw += block[code[c], c]
        return raw / block.max(axis=0).sum()

    def score_batch(self, sequences, strand=1):
        """[P, n] scores of the first W_p bases of every sequence on the GPU: the c_score kernel (cscore.c:174-229).
        NOTE the kernel normalises by the C-style max_raw, which equals the numpy one for every true log-odds PWM."""
        from . import _lib
        vals, widths = self.flat()
        pw, sq = _lib.PwmSet(vals, widths, None), _lib.SeqSet.from_strings(sequences)
        try:
            return _lib.score(pw, sq, strand)
        finally:
            sq.close()
            pw.close()


class Motif:
    """Motif i of a set: the attribute surface the scanner, the writers and the reference's call sites read."""
    __slots__ = ("set", "i")

    def __init__(self, mset, i):
        self.set, self.i = mset, i

    kind = property(lambda self: self.set.kind)
    matrix = property(lambda self: self.set.cols[:, self.set.starts[self.i]:self.set.starts[self.i + 1]])
    length = property(lambda self: int(self.set.widths[self.i]))
    shape = property(lambda self: (4, self.length))
    name = property(lambda self: self.set.names[self.i])
    matrix_id = property(lambda self: self.set.ids[self.i])
    cutoffs = property(lambda self: self.set.cutoffs[self.i])
    max_raw_score = property(lambda self: self.set.raw_extrema()[0][self.i])
    min_raw_score = property(lambda self: self.set.raw_extrema()[1][self.i])

    def __len__(self):
        return self.length

    def __str__(self):
        return "".join(f"{b} {row}\n" for b, row in zip(BASES, self.matrix))

    def _alone(self):
        s = self.set
        return MotifSet(s.kind, self.matrix, [self.length], [self.name], [self.matrix_id], [self.cutoffs], _checked=True)

    def to_ppm(self, normalize=True, pseudo=0.001):
        return self._alone().to_ppm(normalize, pseudo)[0]

    def normalize(self, pseudo=0.001):
        """In place, as the reference's PPM method is (matrix.py:125-147)."""
        self.set.cols[:, self.set.starts[self.i]:self.set.starts[self.i + 1]] = pseudo_normalize(self.matrix, pseudo)

    def to_pwm(self, bg_freq=None):
        return self._alone().to_pwm(bg_freq)[0]

    def set_cutoff(self, p_value, cutoff):
        if self.set.cutoffs[self.i] is None:
            self.set.cutoffs[self.i] = {}
        self.set.cutoffs[self.i][p_value] = cutoff

    def score(self, sequence):
        return self.set.score(self.i, sequence)

    def score_batch(self, sequences, strand=1):
        return self._alone().score_batch(sequences, strand)[0]


def _single(kind, values, name=None, matrix_id=None, cutoffs=None):
    cols = _as_columns(values)
    if kind == "ppm":
        cols = cols.astype(np.float64)                       # normalize() writes in place
    return MotifSet(kind, cols, [cols.shape[1]], [name], [matrix_id], [cutoffs])[0]


# the reference's constructor names (matrix.py:8, 33, 101, 174), as factories of one-motif sets
def PositionMatrix(values, name=None, matrix_id=None):
    return _single("any", values, name, matrix_id)


def PositionFrequencyMatrix(values, name=None, matrix_id=None):
    return _single("pfm", values, name, matrix_id)


def PositionProbabilityMatrix(values, name=None, matrix_id=None):
    return _single("ppm", values, name, matrix_id)


def PositionWeightMatrix(values, name=None, matrix_id=None, cutoffs=None):
    return _single("pwm", values, name, matrix_id, cutoffs)
