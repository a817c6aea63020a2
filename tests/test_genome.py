"""
N3, host side: FASTA -> packed genome -> genome file, against the reference tests' own answers for their toy genome
(tests/golden/ref_genome.json: the data files tests/data/genomes/test/test.fa(.fai) + the literals of tests/test_genome_class.py:14-23 and
tests/test_scanner.py:17,22, made by tests/golden/make_golden.py --only-genome).  No device is touched: the packer is host code
(ms_pack_bases_host), and its planes are held against the oracle's convert_seq.
"""
import json
import os

import numpy as np
import pytest

from motifscan_amd import genome

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(ROOT, "tests", "golden", "ref_genome.json")) as fh:
        return json.load(fh)


@pytest.fixture()
def fasta_dir(gold, tmp_path):
    for name, text in gold["files"].items():
        with open(tmp_path / name, "w", newline="") as fh:
            fh.write(text)
    return tmp_path


def test_fasta_reader_and_packed_genome_give_the_reference_tests_answers(gold, fasta_dir):
    pg = genome.PackedGenome.from_fasta(str(fasta_dir / "test.fa"))              # (with the reference's .fai beside it: checked against it)
    assert pg.names == gold["order"] and pg.chroms == gold["chroms_sorted"] and pg.chrom_sizes == gold["chrom_sizes"]
    for chrom, a, b, want in gold["reference_test_fetches"]:                     # tests/test_genome_class.py:20-23, tests/test_scanner.py:17,22
        assert pg.fetch_sequence(chrom, a, b) == want
    for chrom, cases in gold["all_slices"].items():                              # every 0-based half-open slice, case and N / n kept
        for a, b, want in cases:
            assert pg.fetch_sequence(chrom, a, b) == want
    assert pg.fetch_sequence("chr1", 8, 100) == "Nn" and pg.fetch_sequence("chr1", 5, 5) == ""
    with pytest.raises(KeyError):
        pg.fetch_sequence("chr9", 0, 1)


@pytest.mark.parametrize("name", ["wrapped7.fa", "crlf5.fa"])
def test_multi_line_and_crlf_fasta_read_the_same(gold, fasta_dir, name):
    names, seqs = genome.read_fasta(str(fasta_dir / name))
    assert names == gold["order"]
    assert {n: s.tobytes().decode() for n, s in zip(names, seqs)} == gold["whole"]


def test_a_stale_fai_is_refused_and_bad_fasta_is_reported(fasta_dir):
    with open(fasta_dir / "wrapped7.fa.fai", "w") as fh:
        fh.write("chr1\t11\t6\t10\t11\n")
    with pytest.raises(genome.GenomeFormatError):
        genome.read_fasta(str(fasta_dir / "wrapped7.fa"))
    with open(fasta_dir / "bad.fa", "w") as fh:
        fh.write("ACGT\n>chr1\nAC\n")
    with pytest.raises(genome.GenomeFormatError):
        genome.read_fasta(str(fasta_dir / "bad.fa"))
    with open(fasta_dir / "dup.fa", "w") as fh:
        fh.write(">a\nAC\n>a x\nGT\n")
    with pytest.raises(genome.GenomeFormatError):
        genome.read_fasta(str(fasta_dir / "dup.fa"))
    open(fasta_dir / "empty.fa", "w").close()
    assert genome.read_fasta(str(fasta_dir / "empty.fa")) == ([], [])


def test_genome_file_round_trip_and_damage_is_detected(gold, fasta_dir):
    pg = genome.PackedGenome.from_fasta(str(fasta_dir / "test.fa"))
    path = pg.save(str(fasta_dir / "test.msg"))
    back = genome.PackedGenome.load(path)
    assert back.names == pg.names and back.chrom_sizes == pg.chrom_sizes and np.array_equal(back.offsets, pg.offsets)
    for k in ("codes", "nmask", "lower", "exc_pos", "exc_byte"):
        assert np.array_equal(np.asarray(getattr(back, k)), np.asarray(getattr(pg, k))), k
    for chrom, whole in gold["whole"].items():
        assert back.fetch_sequence(chrom, 0, len(whole)) == whole
    raw = bytearray(open(path, "rb").read())
    raw[-20] ^= 0x10                                                             # one flipped bit in the payload
    with open(fasta_dir / "damaged.msg", "wb") as fh:
        fh.write(raw)
    with pytest.raises(genome.GenomeFormatError):
        genome.PackedGenome.load(str(fasta_dir / "damaged.msg"))
    with open(fasta_dir / "short.msg", "wb") as fh:
        fh.write(raw[:-8])
    with pytest.raises(genome.GenomeFormatError):
        genome.PackedGenome.load(str(fasta_dir / "short.msg"))
    with open(fasta_dir / "other.msg", "wb") as fh:
        fh.write(b"not a genome file" * 8)
    with pytest.raises(genome.GenomeFormatError):
        genome.PackedGenome.load(str(fasta_dir / "other.msg"))


def test_packed_planes_are_convert_seq_of_the_oracle(oracle):
    """The host packer's planes against the oracle's convert_seq (cscore.c:81-114: A/a 0, C/c 1, G/g 2, T/t 3, anything else "no
    contribution") on random bytes of every kind -- IUPAC letters, digits, punctuation, high bytes -- over several chromosomes whose
    boundaries fall inside 32-base units; then the same string through fetch_sequence, byte for byte."""
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"ACGTacgtNnRYKMSWBDHVUrykmswbdhvu-*.0 @`{", dtype=np.uint8)
    p = np.r_[np.full(8, 0.1), np.full(len(alphabet) - 8, 0.2 / (len(alphabet) - 8))]
    seqs = [alphabet[rng.choice(len(alphabet), size=n, p=p)] for n in (1, 31, 32, 33, 1000, 4097, 0, 77)]
    names = [f"c{i}" for i in range(len(seqs))]
    pg = genome.PackedGenome.from_arrays(names, seqs, n_threads=3)
    cat = np.concatenate(seqs)
    idx = oracle.convert_seq(cat.tobytes())                                     # int8: 0..3, -1 = no contribution
    n = cat.size
    units = (n + 31) // 32
    cw = np.asarray(pg.codes).astype(np.uint64)
    cw = cw[0::2] | (cw[1::2] << np.uint64(32))
    code = ((cw[:, None] >> (2 * np.arange(32, dtype=np.uint64))[None, :]) & np.uint64(3)).astype(np.int64).ravel()
    isn = ((np.asarray(pg.nmask)[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(bool).ravel()
    assert np.array_equal(np.where(isn[:n], -1, code[:n]), idx.astype(np.int64))
    assert not code[:n][isn[:n]].any()                                           # a non-ACGT base holds code 0
    assert not code[n:].any() and not isn[n:].any() and code.size == 32 * units  # nothing past the end
    for nm, s in zip(names, seqs):
        assert pg.fetch_sequence(nm, 0, s.size) == s.tobytes().decode("latin-1")
        if s.size > 40:
            assert pg.fetch_sequence(nm, 17, 40) == s[17:40].tobytes().decode("latin-1")
