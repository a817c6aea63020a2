"""
motifscan_amd.matrix -- the log-odds scoring definitions of the reference's
`motifscan.motif.matrix` (/root/reference/motifscan/motif/matrix.py), i.e. what the numbers
fed to the scan kernel mean:

    PositionFrequencyMatrix.to_ppm(normalize, pseudo)      matrix.py:74-98
    PositionProbabilityMatrix.normalize(pseudo)            matrix.py:125-147
    PositionProbabilityMatrix.to_pwm(bg_freq)              matrix.py:149-171   ln(ppm / bg) rounded to 5 decimals
    PositionWeightMatrix.max_raw_score / min_raw_score     matrix.py:202-214   (numpy column extrema, NOT clamped at 0)
    PositionWeightMatrix.score(sequence)                   matrix.py:216-240   scalar forward-strand score
    PositionWeightMatrix.score_batch(sequences, strand)    (new) many sequences at once on the GPU (c_score kernel)

Same constructor checks and the same ValueErrors as the reference classes.
"""
import numpy as np

BASES = "ACGT"


class PositionMatrix:
    """4 x N matrix, rows in the order A, C, G, T."""

    def __init__(self, values, name=None, matrix_id=None):
        if len(values) != 4:
            raise ValueError("values should have exactly 4 rows for A/C/G/T")
        m = np.asarray(values)
        if m.ndim != 2:
            raise ValueError("values should have 2 dimensions in (4 x N)")
        if not (np.issubdtype(m.dtype, np.integer) or np.issubdtype(m.dtype, np.floating)):
            raise ValueError("values should be integers or floating numbers")
        if m.shape[1] == 0:
            raise ValueError("values should have at least 1 position per row")
        self.matrix = m
        self._length = m.shape[1]
        self.name = name
        self.matrix_id = matrix_id

    @property
    def shape(self):
        return self.matrix.shape

    @property
    def length(self):
        return self._length

    def __len__(self):
        return self._length

    def __str__(self):
        return "A {}\nC {}\nG {}\nT {}\n".format(*self.matrix)


class PositionFrequencyMatrix(PositionMatrix):
    def __init__(self, values, name=None, matrix_id=None):
        super().__init__(values, name, matrix_id)
        if not np.issubdtype(self.matrix.dtype, np.integer) or (self.matrix < 0).any():
            raise ValueError("values in PFM should be non-negative integers")
        if (self.matrix.sum(axis=0) == 0).any():
            raise ValueError("all values of a PFM position are 0")

    def to_ppm(self, normalize=True, pseudo=0.001):
        ppm = PositionProbabilityMatrix(self.matrix / self.matrix.sum(axis=0), name=self.name,
                                        matrix_id=self.matrix_id)
        if normalize:
            ppm.normalize(pseudo)
        return ppm


class PositionProbabilityMatrix(PositionMatrix):
    def __init__(self, values, name=None, matrix_id=None):
        super().__init__(values, name, matrix_id)
        if (self.matrix < 0).any():
            raise ValueError("values in PPM should be non-negative numbers")
        col = self.matrix.sum(axis=0)
        if (col == 0).any():
            raise ValueError("all values of a PPM position are 0")
        if not np.allclose(col, 1):
            raise ValueError("the sum probability of a PPM position is not 1")

    def normalize(self, pseudo=0.001):
        """Columns that contain a zero get pseudo/(1-4*pseudo) added to every entry, then every
        column is renormalised to sum 1 ([0,0,10,10] -> [0.001,0.001,0.499,0.499])."""
        if not 0 < pseudo < 0.25:
            raise ValueError("the range of pseudo should be (0, 0.25)")
        bump = pseudo / (1 - 4 * pseudo)
        has_zero = (self.matrix == 0).any(axis=0)
        self.matrix[:, has_zero] += bump
        self.matrix = self.matrix / self.matrix.sum(axis=0)

    def to_pwm(self, bg_freq=None):
        if bg_freq is None:
            bg_freq = {b: 0.25 for b in BASES}
        bg = np.array([bg_freq[b] for b in BASES], dtype=np.float64).reshape(4, 1)
        return PositionWeightMatrix(np.around(np.log(self.matrix / bg), 5), name=self.name,
                                    matrix_id=self.matrix_id)


class PositionWeightMatrix(PositionMatrix):
    def __init__(self, values, name=None, matrix_id=None, cutoffs=None):
        super().__init__(values, name, matrix_id)
        self._max_raw_score = None
        self._min_raw_score = None
        self.cutoffs = cutoffs

    def set_cutoff(self, p_value, cutoff):
        if self.cutoffs is None:
            self.cutoffs = {}
        self.cutoffs[p_value] = cutoff

    @property
    def max_raw_score(self):
        if self._max_raw_score is None:
            self._max_raw_score = self.matrix.max(axis=0).sum()
        return self._max_raw_score

    @property
    def min_raw_score(self):
        if self._min_raw_score is None:
            self._min_raw_score = self.matrix.min(axis=0).sum()
        return self._min_raw_score

    def score(self, sequence):
        """raw / max_raw for one sequence of exactly the PWM's length; non-ACGT letters add nothing."""
        if len(sequence) != self.length:
            raise ValueError("sequence should have the same length as the PWM")
        raw = 0
        for col, nt in enumerate(sequence.upper()):
            row = BASES.find(nt)
            if row >= 0:
                raw += self.matrix[row, col]
        return raw / self.max_raw_score

    def score_batch(self, sequences, strand=1):
        """Scores of many sequences (first `length` bases of each) on the GPU: the c_score kernel
        (cscore.c:174-229).  NOTE the kernel normalises by the C-style max_raw (column maxima
        clamped at 0, cscore.c:39), which equals `max_raw_score` for every true log-odds PWM."""
        from . import _lib
        pw = _lib.PwmSet.from_matrices([self.matrix])
        sq = _lib.SeqSet.from_strings(sequences)
        try:
            return _lib.score(pw, sq, strand)[0]
        finally:
            sq.close()
            pw.close()
