"""
motifscan_amd.formats -- the on-disk formats either side of the scan path:

  in   read_jaspar_pfms(path)                 JASPAR PFM text            motif/__init__.py:70-140
       read_motifscan_pwms(path)              built PWMs + cutoffs       motif/__init__.py:219-319
       write_motifscan_pwms(path, pwms)                                  motif/__init__.py:200-217
  out  write_sites_table(dir, pwms, regions, ...)   motif_sites_number.xls / motif_sites_score.xls  io/__init__.py:12-38
       write_sites_bed(dir, pwms, regions, ...)     motif_sites/<motif>_sites.bed                   io/__init__.py:41-54
       write_enrich_table(dir, results)             motif_enrichment.xls                            io/__init__.py:57-71

The writers take the flat arrays / dense tables the GPU path delivers (no n_pwms x n_regions Python
lists needed) and produce byte-identical files to the reference's writers given identical hits:
numbers are printed with Python's `str()` exactly as the reference does (io/__init__.py:34,54).
The same strictness as the reference's parsers: anything that is not a header / matrix / cutoff
line in the expected order raises, with the 1-based line number.
"""
import os
import re

import numpy as np

from .matrix import BASES, PositionFrequencyMatrix, PositionWeightMatrix


class PfmsJasparFormatError(Exception):
    def __init__(self, line_num, line):
        super().__init__(f"Invalid JASPAR PFMs format at line {line_num}: {line!r}")
        self.line_num = line_num


class PwmsMotifScanFormatError(Exception):
    def __init__(self, line_num, line):
        super().__init__(f"Invalid MotifScan PWMs format at line {line_num}: {line!r}")
        self.line_num = line_num


_JASPAR_HEADER = re.compile(r"^>\s*(\S+)(\s+(\S+))?")
_JASPAR_ROW = re.compile(r"\s*([ACGT])\s*\[\s*(.+)\s*\]")
_PWM_HEADER = re.compile(r"^>(\S+)\t(\S+)\tPWM$")
_PWM_ROW = re.compile(r"^([ACGT]) \[(.+)\]$")
_PWM_CUTOFF = re.compile(r"^Cutoff_p(\S+)\t(\S+)")


def _numbered_lines(path):
    """(1-based line number, stripped text) of every non-blank line, then (lines + 1, None)."""
    total = 0
    with open(path, "r") as fh:
        for total, line in enumerate(fh, 1):
            line = line.strip()
            if line:
                yield total, line
    yield total + 1, None


def read_jaspar_pfms(path):
    """JASPAR PFMs: a '>id name' header followed by exactly four rows in A, C, G, T order, either
    'A [ 3 0 ... ]' or bare numbers.  Returns a list of PositionFrequencyMatrix."""
    pfms, rows, meta = [], None, None
    last = 0
    for num, line in _numbered_lines(path):
        if line is None:
            last = num
            break
        header = _JASPAR_HEADER.match(line)
        if rows is None:                                    # a header must open every matrix
            if not header:
                raise PfmsJasparFormatError(num, line)
            meta, rows = (header.group(1), header.group(3)), []
            continue
        if header:
            raise PfmsJasparFormatError(num, line)
        m = _JASPAR_ROW.match(line)
        if m:
            if m.group(1) != BASES[len(rows)]:
                raise PfmsJasparFormatError(num, line)
            fields = m.group(2).split()
        else:
            fields = line.split()
        try:
            rows.append([int(x) for x in fields])
        except (ValueError, TypeError):
            raise PfmsJasparFormatError(num, line)
        if len(rows) == 4:
            pfms.append(PositionFrequencyMatrix(rows, name=meta[1], matrix_id=meta[0]))
            rows = None
    if rows is not None:
        raise PfmsJasparFormatError(last, "")
    return pfms


def read_motifscan_pwms(path):
    """MotifScan built PWMs: '>id<TAB>name<TAB>PWM', four 'A [..]' rows in order, then >= 1
    'Cutoff_p<p><TAB><value>' lines.  Returns a list of PositionWeightMatrix with .cutoffs."""
    pwms = []
    state = "header"                                        # header -> rows -> cutoff -> cutoff_or_header
    meta, rows, cutoffs = None, [], {}
    last = 0

    def flush():
        pwms.append(PositionWeightMatrix(rows, name=meta[1], matrix_id=meta[0], cutoffs=cutoffs))

    for num, line in _numbered_lines(path):
        if line is None:
            last = num
            break
        header, row, cut = _PWM_HEADER.match(line), _PWM_ROW.match(line), _PWM_CUTOFF.match(line)
        if header:
            if state not in ("header", "cutoff_or_header"):
                raise PwmsMotifScanFormatError(num, line)
            if state == "cutoff_or_header":
                flush()
            meta, rows, cutoffs, state = (header.group(1), header.group(2)), [], {}, "rows"
        elif row:
            if state != "rows" or row.group(1) != BASES[len(rows)]:
                raise PwmsMotifScanFormatError(num, line)
            try:
                rows.append([float(x) for x in row.group(2).split()])
            except (ValueError, TypeError):
                raise PwmsMotifScanFormatError(num, line)
            if len(rows) == 4:
                state = "cutoff"
        elif cut:
            if state not in ("cutoff", "cutoff_or_header"):
                raise PwmsMotifScanFormatError(num, line)
            cutoffs[cut.group(1)] = float(cut.group(2))
            state = "cutoff_or_header"
        else:
            raise PwmsMotifScanFormatError(num, line)
    if state in ("rows", "cutoff"):
        raise PwmsMotifScanFormatError(last, "")
    if state == "cutoff_or_header":
        flush()
    return pwms


def write_motifscan_pwms(path, pwms):
    with open(path, "w") as out:
        for pwm in pwms:
            out.write(f">{pwm.matrix_id}\t{pwm.name}\tPWM\n")
            for base, row in zip(BASES, pwm.matrix):
                out.write(base + " [" + "\t".join(f"{v:8.5f}" for v in row) + "]\n")
            for p, cutoff in pwm.cutoffs.items():
                out.write(f"Cutoff_p{p}\t{cutoff}\n")


# ------------------------------------------------------------------------ result writers --

def _motif_label(pwm):
    return pwm.matrix_id + "," + pwm.name


def write_sites_table(output_dir, pwms, regions, n_sites, max_score):
    """n_sites int [P][R], max_score float [P][R] (NaN = no site -> 'NA'): the dense tables of
    ScanResult.site_tables / Scanner.scan_motifs_arrays(with_tables=True)."""
    os.makedirs(output_dir, exist_ok=True)
    header = "chr\tstart\tend\t" + "\t".join(_motif_label(p) for p in pwms) + "\n"
    n_sites = np.asarray(n_sites)
    max_score = np.asarray(max_score, dtype=np.float64)
    with open(os.path.join(output_dir, "motif_sites_number.xls"), "w") as f_num, \
            open(os.path.join(output_dir, "motif_sites_score.xls"), "w") as f_score:
        f_num.write(header)
        f_score.write(header)
        for r, region in enumerate(regions):
            lead = f"{region.chrom}\t{region.start + 1}\t{region.end}\t"
            counts = n_sites[:, r].tolist()
            scores = max_score[:, r].tolist()
            f_num.write(lead + "\t".join(str(c) for c in counts) + "\n")
            f_score.write(lead + "\t".join("NA" if c == 0 else str(s) for c, s in zip(counts, scores)) + "\n")


def write_sites_bed(output_dir, pwms, regions, hits):
    """hits: the flat dict of Scanner.scan_motifs_arrays (motif, region, start, score, strand,
    motif_offsets) -- one '<id>_<name>_sites.bed' per motif, sites in scan order."""
    out_dir = os.path.join(output_dir, "motif_sites")
    os.makedirs(out_dir, exist_ok=True)
    off = np.asarray(hits["motif_offsets"])
    region, start, score, strand = (np.asarray(hits[k]).tolist() for k in ("region", "start", "score", "strand"))
    for m, pwm in enumerate(pwms):
        name = re.sub("[-:./*]", "_", pwm.matrix_id + "_" + pwm.name)
        with open(os.path.join(out_dir, f"{name}_sites.bed"), "w") as out:
            for k in range(int(off[m]), int(off[m + 1])):
                out.write(f"{regions[region[k]].chrom}\t{start[k]}\t{start[k] + pwm.length}\t.\t{score[k]}\t"
                          f"{'+' if strand[k] == 1 else '-'}\n")


def write_enrich_table(output_dir, names, rows):
    """rows: motifscan_amd.dist.enrichment() tuples (n_input, n_control, fold, p_enriched, p_depleted,
    p_corrected); sorted by enriched p-value, then larger fold change first (io/__init__.py:63)."""
    os.makedirs(output_dir, exist_ok=True)
    order = sorted(range(len(rows)), key=lambda i: (rows[i][3], -rows[i][2]))
    with open(os.path.join(output_dir, "motif_enrichment.xls"), "w") as out:
        out.write("Motif\tNum_input_regions\tNum_control_regions\tFold_change\tEnriched_P_value\tDepleted_P_value\t"
                  "Corrected_P_value\n")
        for i in order:
            a, c, fold, pe, pd, pc = rows[i]
            out.write(f"{names[i]}\t{a}\t{c}\t{fold}\t{pe}\t{pd}\t{pc}\n")
