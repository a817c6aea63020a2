"""
motifscan_amd.genome -- FASTA -> packed genome file -> genome resident in HBM ("pack the FASTA once").

What it replaces on the measured path: `Genome.__init__` opening the FASTA through pysam / htslib and one `fetch_sequence` call per
region (/root/reference/motifscan/genome/__init__.py:61-83, 117-135; scanner.py:71-87).  Here the FASTA is read ONCE by a plain reader
(no pysam; an optional `.fai` beside it is used to check the chromosome table), converted with convert_seq's rules (cscore.c:81-114) into
the two planes of the device layout -- 2-bit codes + a non-ACGT bit per base, 0.375 B/base -- by the library's host packer
(`ms_pack_bases_host`), and written to a genome file; every later process maps that file and uploads the planes as they are
(`ms_genome_create_packed`: one host-to-device copy of 1.1 GB for 3 Gbp, no ASCII, no pack kernel).

`PackedGenome` quacks like the part of the reference's `Genome` the scan path reads -- `chroms` (sorted, genome/__init__.py:88-99),
`chrom_sizes`, `fetch_sequence(chrom, start, end)` (0-based, half-open, the reference tests' literals: tests/test_genome_class.py:14-23) --
and `fetch_sequence` returns the FASTA's own bytes, case and IUPAC letters included: besides the two planes the file keeps a soft-mask
(case) plane and the (rare) non-ACGT bytes that are not N / n, so that a string cut from the file equals a string cut from the FASTA
(scanning never reads them: lower case = upper case, every non-ACGT byte "adds nothing").

File layout (little endian), version 1:
    0    8  magic "MSGENOM1"
    8    4  uint32 version (1)
    12   4  uint32 n_chroms
    16   8  int64  n_bases
    24   8  int64  n_exceptions
    32   4  uint32 crc32 of the four payload sections
    36   4  uint32 flags (bit 0: case plane present)
    40  24  reserved (0)
    64      int64 offsets[n_chroms + 1]; then per chromosome uint16 name length + UTF-8 name; zero padding to a multiple of 64
    ...     codes   uint32[2 * units]      units = ceil(n_bases / 32)
    ...     nmask   uint32[units]
    ...     lower   uint32[units]          (bit i: base i is a lower-case letter)     -- if flags bit 0
    ...     exceptions: int64 pos[n_exceptions], then uint8 byte[n_exceptions]       (non-ACGT bytes other than N / n, by position)
"""
import os
import struct
import zlib

import numpy as np

MAGIC = b"MSGENOM1"
VERSION = 1
_HDR = struct.Struct("<8sIIqqII24x")
assert _HDR.size == 64


class GenomeFormatError(ValueError):
    pass


# --------------------------------------------------------------------------- FASTA --

def read_fasta(path):
    """Plain FASTA reader -> (names, [uint8 array per record]) in file order.  A record's name is the header up to the first white
    space (what htslib / faidx index by); sequence lines are joined, line ends ('\\n', '\\r\\n') and blank lines dropped, every other byte kept
    as it is (case, N, IUPAC).  Vectorised: the file is read once and never walked in Python.
    If `path + '.fai'` exists its names and lengths must agree (GenomeFormatError otherwise): a stale index beside a replaced FASTA is the
    usual way a genome directory goes wrong."""
    buf = np.fromfile(path, dtype=np.uint8)
    n = buf.size
    if n == 0:
        return [], []
    nl = np.flatnonzero(buf == 10)
    line_start = np.concatenate([[0], nl + 1])
    line_start = line_start[line_start < n]
    hdr_lines = line_start[buf[line_start] == ord(">")]
    lead = buf[:hdr_lines[0]] if hdr_lines.size else buf
    if np.any((lead != 10) & (lead != 13)):
        raise GenomeFormatError(f"{path}: sequence data before the first '>' header")
    if hdr_lines.size == 0:
        return [], []
    names, seqs = [], []
    ends = np.concatenate([hdr_lines[1:], [n]])
    for h, e in zip(hdr_lines.tolist(), ends.tolist()):
        k = int(np.searchsorted(nl, h))                              # the header's own line end
        h_end = int(nl[k]) if k < nl.size and nl[k] < e else e
        header = buf[h + 1:h_end].tobytes().decode("utf-8", "replace").strip()
        if not header:
            raise GenomeFormatError(f"{path}: empty header at byte {h}")
        names.append(header.split()[0])
        body = buf[min(h_end + 1, e):e]
        seqs.append(np.ascontiguousarray(body[(body != 10) & (body != 13)]))
    if len(set(names)) != len(names):
        raise GenomeFormatError(f"{path}: duplicate sequence names")
    fai = path + ".fai"
    if os.path.isfile(fai):
        want = []
        with open(fai) as fh:
            for line in fh:
                f = line.rstrip("\n").split("\t")
                if len(f) >= 2:
                    want.append((f[0], int(f[1])))
        got = [(nm, int(s.size)) for nm, s in zip(names, seqs)]
        if want != got:
            raise GenomeFormatError(f"{fai} does not describe {path} (index: {want[:3]}..., file: {got[:3]}...): rebuild the index")
    return names, seqs


# --------------------------------------------------------------------------- packed genome --

def _pack_host(bases, n_threads):
    import ctypes
    from . import _lib
    n = int(bases.size)
    units = (n + 31) // 32
    codes, nmask = np.zeros(2 * units, dtype=np.uint32), np.zeros(units, dtype=np.uint32)
    _lib.check(_lib.lib().ms_pack_bases_host(bases.ctypes.data_as(ctypes.c_char_p), n, int(n_threads),
                                             _lib.ptr(codes, ctypes.c_uint32), _lib.ptr(nmask, ctypes.c_uint32)))
    return codes, nmask


def _bits_to_words(bits, units):
    """bool[n] -> uint32[units], bit i of word u = bits[32 u + i]"""
    packed = np.packbits(bits, bitorder="little")
    out = np.zeros(units * 4, dtype=np.uint8)
    out[:packed.size] = packed
    return out.view("<u4").astype(np.uint32, copy=False)


class PackedGenome:
    """A genome in the device layout on the HOST (arrays, or memory maps of a genome file)."""

    def __init__(self, names, offsets, codes, nmask, lower=None, exc_pos=None, exc_byte=None):
        self.names = list(names)
        self.index = {nm: i for i, nm in enumerate(self.names)}
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.codes, self.nmask, self.lower = codes, nmask, lower
        self.exc_pos = np.zeros(0, dtype=np.int64) if exc_pos is None else exc_pos
        self.exc_byte = np.zeros(0, dtype=np.uint8) if exc_byte is None else exc_byte
        self.n_bases = int(self.offsets[-1]) if self.offsets.size else 0
        self.chrom_sizes = {nm: int(self.offsets[i + 1] - self.offsets[i]) for i, nm in enumerate(self.names)}

    # ---- the reference Genome's surface (genome/__init__.py:88-135) ----
    @property
    def chroms(self):
        return sorted(self.names)

    def fetch_sequence(self, chrom, start, end):
        """The FASTA's bytes of chrom[start:end) (0-based, half-open; clipped like a Python slice / pysam's fetch of an in-range
        request), decoded from the planes: ACGT from the codes, case from the soft-mask plane, N / n where the non-ACGT bit is set,
        any other letter from the exception list."""
        i = self.index[chrom]
        size = self.chrom_sizes[chrom]
        start, end = max(0, int(start)), min(int(end), size)
        if end <= start:
            return ""
        g0, g1 = int(self.offsets[i]) + start, int(self.offsets[i]) + end
        u0, u1 = g0 // 32, (g1 + 31) // 32
        cw = np.asarray(self.codes[2 * u0:2 * u1]).astype(np.uint64)
        cw = cw[0::2] | (cw[1::2] << np.uint64(32))
        sh = (2 * np.arange(32, dtype=np.uint64))[None, :]
        code = ((cw[:, None] >> sh) & np.uint64(3)).astype(np.uint8).ravel()[g0 - 32 * u0:g1 - 32 * u0]
        bit = np.arange(32, dtype=np.uint32)[None, :]
        isn = ((np.asarray(self.nmask[u0:u1])[:, None] >> bit) & 1).astype(bool).ravel()[g0 - 32 * u0:g1 - 32 * u0]
        out = np.frombuffer(b"ACGT", dtype=np.uint8)[code].copy()
        out[isn] = ord("N")
        if self.lower is not None:
            low = ((np.asarray(self.lower[u0:u1])[:, None] >> bit) & 1).astype(bool).ravel()[g0 - 32 * u0:g1 - 32 * u0]
            out[low] |= 0x20
        if self.exc_pos.size:
            a, b = np.searchsorted(self.exc_pos, [g0, g1])
            if b > a:
                out[np.asarray(self.exc_pos[a:b]) - g0] = np.asarray(self.exc_byte[a:b])
        return out.tobytes().decode("latin-1")

    # ---- construction ----
    @classmethod
    def from_arrays(cls, names, seqs, n_threads=None, keep_case=True):
        """names + one uint8 array (or bytes / str) per chromosome -> the planes, packed by the library's host packer."""
        arrs = [np.frombuffer(s.encode("latin-1") if isinstance(s, str) else bytes(s), dtype=np.uint8) if not isinstance(s, np.ndarray)
                else np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
        offsets = np.zeros(len(arrs) + 1, dtype=np.int64)
        if arrs:
            offsets[1:] = np.cumsum([a.size for a in arrs])
        bases = np.concatenate(arrs) if arrs else np.zeros(0, dtype=np.uint8)
        n = int(bases.size)
        units = (n + 31) // 32
        codes, nmask = _pack_host(bases, n_threads or min(16, os.cpu_count() or 1))
        lower = exc_pos = exc_byte = None
        if keep_case and n:
            fold = bases | 0x20
            letter = (fold >= ord("a")) & (fold <= ord("z"))
            lower = _bits_to_words(letter & ((bases & 0x20) != 0), units)
            acgt = (fold == ord("a")) | (fold == ord("c")) | (fold == ord("g")) | (fold == ord("t"))
            odd = np.flatnonzero(~acgt & (fold != ord("n")))            # non-ACGT bytes that do not read back as N / n
            exc_pos, exc_byte = odd.astype(np.int64), bases[odd].copy()
        return cls(names, offsets, codes, nmask, lower, exc_pos, exc_byte)

    @classmethod
    def from_fasta(cls, path, n_threads=None, keep_case=True):
        names, seqs = read_fasta(path)
        return cls.from_arrays(names, seqs, n_threads=n_threads, keep_case=keep_case)

    # ---- the genome file ----
    def _payload(self):
        parts = [np.ascontiguousarray(self.codes, dtype="<u4"), np.ascontiguousarray(self.nmask, dtype="<u4")]
        if self.lower is not None:
            parts.append(np.ascontiguousarray(self.lower, dtype="<u4"))
        parts += [np.ascontiguousarray(self.exc_pos, dtype="<i8"), np.ascontiguousarray(self.exc_byte, dtype=np.uint8)]
        return parts

    def save(self, path):
        parts = self._payload()
        crc = 0
        for p in parts:
            crc = zlib.crc32(memoryview(p).cast("B"), crc)
        table = self.offsets.astype("<i8").tobytes()
        for nm in self.names:
            raw = nm.encode("utf-8")
            if len(raw) > 0xFFFF:
                raise ValueError("chromosome name too long")
            table += struct.pack("<H", len(raw)) + raw
        table += b"\0" * (-len(table) % 64)
        tmp = path + ".tmp"
        with open(tmp, "wb") as fh:
            fh.write(_HDR.pack(MAGIC, VERSION, len(self.names), self.n_bases, int(self.exc_pos.size), crc & 0xFFFFFFFF,
                               1 if self.lower is not None else 0))
            fh.write(table)
            for p in parts:
                fh.write(memoryview(p).cast("B"))
        os.replace(tmp, path)
        return path

    @classmethod
    def load(cls, path, verify=True):
        """Memory-maps the file (nothing is read until it is uploaded or fetched from).  verify: check the payload's CRC-32 -- one
        sequential read of the file, ~1 s per GB; the upload validates the planes' invariants either way (ms_genome_create_packed)."""
        size = os.path.getsize(path)
        with open(path, "rb") as fh:
            head = fh.read(64)
            if len(head) < 64 or head[:8] != MAGIC:
                raise GenomeFormatError(f"{path} is not a motifscan_amd genome file")
            _, version, n_chroms, n_bases, n_exc, crc, flags = _HDR.unpack(head)
            if version != VERSION:
                raise GenomeFormatError(f"{path}: genome file version {version}, this build reads {VERSION}")
            if n_bases < 0 or n_exc < 0 or n_chroms > (1 << 24):
                raise GenomeFormatError(f"{path}: corrupt header")
            offsets = np.frombuffer(fh.read(8 * (n_chroms + 1)), dtype="<i8").astype(np.int64)
            if offsets.size != n_chroms + 1 or offsets[0] != 0 or np.any(np.diff(offsets) < 0) or offsets[-1] != n_bases:
                raise GenomeFormatError(f"{path}: corrupt chromosome table")
            names = []
            for _ in range(n_chroms):
                raw = fh.read(2)
                if len(raw) < 2:
                    raise GenomeFormatError(f"{path}: truncated chromosome table")
                (ln,) = struct.unpack("<H", raw)
                names.append(fh.read(ln).decode("utf-8"))
            pos = fh.tell()
        pos += -pos % 64
        units = (n_bases + 31) // 32
        has_lower = bool(flags & 1)
        need = pos + 4 * units * (3 + (1 if has_lower else 0)) + 9 * n_exc
        if size != need:
            raise GenomeFormatError(f"{path}: {size} bytes on disk, the header describes {need}")

        def section(dtype, count):
            nonlocal pos
            a = np.memmap(path, dtype=dtype, mode="r", offset=pos, shape=(count,)) if count else np.zeros(0, dtype=dtype)
            pos += count * np.dtype(dtype).itemsize
            return a
        codes, nmask = section("<u4", 2 * units), section("<u4", units)
        lower = section("<u4", units) if has_lower else None
        exc_pos, exc_byte = section("<i8", n_exc), section(np.uint8, n_exc)
        g = cls(names, offsets, codes, nmask, lower, exc_pos, exc_byte)
        if verify:
            c = 0
            for p in g._payload():
                c = zlib.crc32(memoryview(np.ascontiguousarray(p)).cast("B"), c)
            if (c & 0xFFFFFFFF) != crc:
                raise GenomeFormatError(f"{path}: checksum mismatch (file damaged)")
        return g

    def to_resident(self):
        """Upload the planes: the genome resident in HBM (a `_lib.ResidentGenome` whose host side is this object)."""
        from . import _lib
        return _lib.ResidentGenome.from_packed(self)
