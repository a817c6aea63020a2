"""
N4: the on-disk formats either side of the scan path, against what the REAL reference reads and
writes (tests/golden/ref_small.json["N4"], produced by tests/golden/make_golden.py).  CPU only.
"""
import os

import numpy as np
import pytest

from motifscan_amd import dist, formats, matrix


@pytest.fixture()
def n4(small, tmp_path):
    d = small["N4"]
    for rel, text in d["files"].items():
        path = tmp_path / rel
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_text(text)
    return d, tmp_path


def test_read_reference_data_files(n4):
    d, root = n4
    pfms = formats.read_jaspar_pfms(root / "test/test_pfms.jaspar")
    assert [(p.matrix_id, p.name, p.matrix.tolist()) for p in pfms] == \
        [(e["matrix_id"], e["name"], e["matrix"]) for e in d["pfms"]]
    pwms = formats.read_motifscan_pwms(root / "test/test_pwms.motifscan")
    assert [(p.matrix_id, p.name, p.matrix.tolist(), p.cutoffs) for p in pwms] == \
        [(e["matrix_id"], e["name"], e["matrix"], e["cutoffs"]) for e in d["pwms"]]
    assert [p.length for p in pwms] == [6, 17]


def test_malformed_files_fail_like_the_reference(n4):
    d, root = n4
    assert len(d["bad"]) == 10 and all(d["bad"].values())
    for rel, message in d["bad"].items():
        reader, exc = ((formats.read_jaspar_pfms, formats.PfmsJasparFormatError) if rel.endswith(".jaspar")
                       else (formats.read_motifscan_pwms, formats.PwmsMotifScanFormatError))
        with pytest.raises(exc) as e:
            reader(root / rel)
        assert str(e.value) == message, rel


def test_write_motifscan_pwms_byte_identical(n4):
    d, root = n4
    pwms = formats.read_motifscan_pwms(root / "test/test_pwms.motifscan")
    built = matrix.PositionFrequencyMatrix(d["pfms"][1]["matrix"], name="Alx1", matrix_id="MA0854.1").to_ppm().to_pwm(
        {"A": 0.3, "C": 0.3, "G": 0.15, "T": 0.25})
    assert built.matrix.tolist() == d["built_matrix"]
    built.set_cutoff("1e-3", 0.20892548)
    built.set_cutoff("1e-4", float(np.around(0.46693615340298805, 8)))
    out = root / "w.motifscan"
    formats.write_motifscan_pwms(out, pwms + [built])
    assert out.read_text() == d["written_pwms"]
    again = formats.read_motifscan_pwms(out)                         # round trip
    assert again[2].cutoffs == built.cutoffs and np.array_equal(again[2].matrix, built.matrix)


def test_result_writers_byte_identical(oracle, small, n4):
    """hits (here from the oracle; on the GPU box from the device) -> flat arrays / dense tables ->
    the three result files, byte for byte what the reference's io module wrote."""
    d, root = n4
    w = d["writers"]
    chroms = small["G2"]["chroms"]
    pwms = formats.read_motifscan_pwms(root / "test/test_pwms.motifscan")

    class Reg:
        def __init__(self, c, s, e):
            self.chrom, self.start, self.end = c, s, e

    def scan(region_rows):
        regs = [Reg(*r) for r in region_rows]
        seqs = [chroms[r.chrom][r.start:r.end] for r in regs]
        sites = oracle.c_scan_motif([p.matrix.tolist() for p in pwms], [p.cutoffs[w["p_value"]] for p in pwms], seqs, 3, 1)
        ms = oracle.deduplicate_motif_sites(oracle.make_motif_sites(sites, [r.start for r in regs]), [p.length for p in pwms])
        rows = [(m, r, s.start, s.score, 1 if s.strand == "+" else 2) for m, per in enumerate(ms) for r, ss in enumerate(per) for s in ss]
        n_sites = np.array([[len(x) for x in per] for per in ms])
        max_score = np.array([[max(s.score for s in x) if x else np.nan for x in per] for per in ms])
        hits = {"motif": np.array([x[0] for x in rows], dtype=np.int64), "region": [x[1] for x in rows],
                "start": [x[2] for x in rows], "score": [x[3] for x in rows], "strand": [x[4] for x in rows]}
        hits["motif_offsets"] = np.concatenate([[0], np.cumsum(np.bincount(hits["motif"], minlength=len(pwms)))])
        return regs, hits, n_sites, max_score

    regs, hits, n_sites, max_score = scan(w["regions"])
    out = root / "out"
    formats.write_sites_table(out, pwms, regs, n_sites, max_score)
    formats.write_sites_bed(out, pwms, regs, hits)
    assert (out / "motif_sites_number.xls").read_text() == w["motif_sites_number.xls"]
    assert (out / "motif_sites_score.xls").read_text() == w["motif_sites_score.xls"]
    assert {f: (out / "motif_sites" / f).read_text() for f in os.listdir(out / "motif_sites")} == w["bed"]
    cregs, _, cn, _ = scan(w["control_regions"])
    rows = dist.enrichment((n_sites > 0).sum(axis=1), (cn > 0).sum(axis=1), len(regs), len(cregs))
    formats.write_enrich_table(out, [p.matrix_id + "," + p.name for p in pwms], rows)
    assert (out / "motif_enrichment.xls").read_text() == w["motif_enrichment.xls"]
