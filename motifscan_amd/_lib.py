"""
motifscan_amd._lib -- ctypes binding of libmotifscan_amd.so (include/motifscan_amd.h).

The shared library is built in-tree by motifscan_amd/csrc/Makefile (hipcc, gfx950).  There is
no CPU implementation behind it: if the library is missing, or no MI355X is visible when a
compute entry point is called, the call raises -- it never falls back.
"""
import ctypes
import importlib.util
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MS_LIB_VARIANT=asm loads the variant whose pre-filter holds the hand-written gfx950 blocks of rounds 4-5 (csrc/Makefile, ms_kernels.hip
# "the two builds"): same C-ABI, same results, same speed as measured in round 6 -- kept for A/B runs; the GPU suite runs the goldens on both.
LIB_VARIANT = os.environ.get("MS_LIB_VARIANT", "")
if LIB_VARIANT not in ("", "asm"):
    raise RuntimeError(f"MS_LIB_VARIANT={LIB_VARIANT!r}: known variants are '' (default) and 'asm'")
LIB_PATH = os.path.join(_HERE, "libmotifscan_amd_asm.so" if LIB_VARIANT == "asm" else "libmotifscan_amd.so")

MS_OK, MS_ERR_INVALID, MS_ERR_NOMEM, MS_ERR_RUNTIME = 0, 1, 2, 3
MS_SCAN_DEFAULT, MS_SCAN_EXACT_ONLY, MS_SCAN_COUNTS_ONLY = 0, 1, 2
MS_STREAM_DEDUP, MS_STREAM_NO_HITS, MS_STREAM_EXACT_ONLY, MS_STREAM_PACKED, MS_STREAM_HOST_PACK, MS_STREAM_PACKED12 = 1, 2, 4, 8, 16, 32


class ScanStats(ctypes.Structure):
    _fields_ = [("n_bases", ctypes.c_int64), ("n_windows", ctypes.c_int64), ("n_candidates", ctypes.c_int64),
                ("n_hits", ctypes.c_int64), ("n_pwms", ctypes.c_int32), ("n_pwms_exact", ctypes.c_int32),
                ("n_tiles", ctypes.c_int32), ("n_passes", ctypes.c_int32), ("ms_prefilter", ctypes.c_double),
                ("ms_exact", ctypes.c_double), ("ms_sort", ctypes.c_double), ("ms_finalize", ctypes.c_double),
                ("ms_total", ctypes.c_double), ("lds_bytes_read", ctypes.c_int64),
                ("hbm_bytes_algorithmic", ctypes.c_int64), ("pf_clock_mhz", ctypes.c_double),
                ("mfma_ops", ctypes.c_int64), ("mfma_ops_algorithmic", ctypes.c_int64), ("pf_engine", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Span(ctypes.Structure):
    """ms_span: one span of a host-streamed window sweep."""
    _fields_ = [("chrom", ctypes.c_int32), ("reserved", ctypes.c_int32), ("begin", ctypes.c_int64), ("end", ctypes.c_int64),
                ("first_window", ctypes.c_int64), ("n_windows", ctypes.c_int64)]


_lib = None


def build(verbose=False):
    """Compile libmotifscan_amd.so in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.run(cmd, check=True)
    return LIB_PATH


def _preload_hip_runtime():
    """torch wheels carry their own libamdhip64.so (same SONAME as /opt/rocm's).  If torch is
    installed, map ITS copy first so that this library and torch share one HIP runtime in the
    process whichever gets imported first."""
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is not built (run `python -c 'import __graft_entry__ as g; g.build()'` or "
            f"`make -C motifscan_amd/csrc`); motifscan_amd has no CPU fallback")
    _preload_hip_runtime()
    L = ctypes.CDLL(LIB_PATH)
    c_int, c_i32, c_i64, c_u32, vp = ctypes.c_int, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32, ctypes.c_void_p
    pd, pi32, pi64 = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i32), ctypes.POINTER(c_i64)
    pi8, pu8, pu32 = ctypes.POINTER(ctypes.c_int8), ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(c_u32)
    pvp = ctypes.POINTER(vp)
    sig = {
        "ms_last_error": (ctypes.c_char_p, []),
        "ms_version": (c_int, []),
        "ms_build_flags": (c_int, []),
        "ms_device_count": (c_int, [ctypes.POINTER(c_int)]),
        "ms_set_device": (c_int, [c_int]),
        "ms_device_name": (c_int, [ctypes.c_char_p, c_int]),
        "ms_numa_bind_thread": (c_int, [c_int, ctypes.POINTER(c_int)]),
        "ms_pwmset_create": (c_int, [pd, pi32, pd, c_i32, pvp]),
        "ms_pwmset_set_cutoffs": (c_int, [vp, pd]),
        "ms_pwmset_size": (c_int, [vp, pi32]),
        "ms_pwmset_max_raw": (c_int, [vp, pd]),
        "ms_pwmset_free": (None, [vp]),
        "ms_seqset_create": (c_int, [ctypes.c_char_p, pi64, c_i64, c_int, pvp]),
        "ms_seqset_create_hostpacked": (c_int, [ctypes.c_char_p, pi64, c_i64, c_int, pvp]),
        "ms_debug_host_pack": (c_int, [ctypes.c_char_p, pi64, c_i64, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), pi32, pi32]),
        "ms_seqset_from_device": (c_int, [vp, pi64, c_i64, pvp]),
        "ms_seqset_repack": (c_int, [vp]),
        "ms_seqset_size": (c_int, [vp, pi64, pi64]),
        "ms_seqset_free": (None, [vp]),
        "ms_genome_create": (c_int, [ctypes.c_char_p, pi64, c_i32, pvp]),
        "ms_genome_size": (c_int, [vp, pi32, pi64]),
        "ms_genome_create_packed": (c_int, [pu32, pu32, pi64, c_i32, pvp]),
        "ms_genome_packed_host": (c_int, [vp, pu32, pu32]),
        "ms_pack_bases_host": (c_int, [ctypes.c_char_p, c_i64, c_int, pu32, pu32]),
        "ms_genome_free": (None, [vp]),
        "ms_seqset_from_genome": (c_int, [vp, pi32, pi64, pi64, c_i64, pvp]),
        "ms_scan": (c_int, [vp, vp, c_int, c_u32, pvp]),
        "ms_scan_sweep": (c_int, [vp, vp, c_i32, c_i64, c_i64, c_i32, c_i32, c_int, c_u32, pvp]),
        "ms_scan_regions_once": (c_int, [vp, vp, pi32, pi64, pi64, c_i64, c_int, c_u32, pvp]),
        "ms_result_num_hits": (c_int, [vp, pi64]),
        "ms_result_motif_offsets": (c_int, [vp, pi64]),
        "ms_result_hits": (c_int, [vp, pi64, pi64, pd, pi8]),
        "ms_result_hits_host": (c_int, [vp, ctypes.POINTER(pi64), ctypes.POINTER(pi64), ctypes.POINTER(pd), ctypes.POINTER(pi8)]),
        "ms_result_hits_packed_host": (c_int, [vp, ctypes.POINTER(ctypes.POINTER(ctypes.c_uint64)), ctypes.POINTER(pd)]),
        "ms_result_hits_packed12_host": (c_int, [vp, ctypes.POINTER(pu32), ctypes.POINTER(pd), pi32]),
        "ms_result_packed_form": (c_int, [vp, pi32]),
        "ms_host_alloc": (c_int, [ctypes.c_size_t, pvp]),
        "ms_device_pool_stats": (c_int, [ctypes.POINTER(ctypes.c_uint64)]),
        "ms_host_free": (None, [vp]),
        "ms_host_pool_stats": (c_int, [ctypes.POINTER(ctypes.c_uint64)]),
        "ms_stream_create": (c_int, [vp, c_int, c_u32, c_int, pvp]),
        "ms_stream_submit": (c_int, [vp, vp, pi64, c_i64]),
        "ms_stream_submit_counts_only": (c_int, [vp, vp, pi64, c_i64]),
        "ms_stream_submit_span": (c_int, [vp, vp, c_i64, c_i32, c_i32]),
        "ms_stream_submit_regions": (c_int, [vp, vp, pi32, pi64, pi64, c_i64]),
        "ms_stream_next": (c_int, [vp, pvp]),
        "ms_stream_in_flight": (c_int, [vp, ctypes.POINTER(c_int)]),
        "ms_stream_capacity": (c_int, [vp, ctypes.POINTER(c_int)]),
        "ms_stream_stats": (c_int, [vp, ctypes.POINTER(ctypes.c_double)]),
        "ms_stream_free": (None, [vp]),
        "ms_sweep_spans": (c_int, [pi64, c_i32, c_i32, c_i32, c_i64, ctypes.POINTER(Span), c_i64, pi64]),
        "ms_result_region_counts": (c_int, [vp, pi64]),
        "ms_result_region_counts_device": (c_int, [vp, pvp]),
        "ms_result_stats": (c_int, [vp, ctypes.POINTER(ScanStats)]),
        "ms_result_dedup": (c_int, [vp, vp]),
        "ms_result_site_tables": (c_int, [vp, pi32, pd]),
        "ms_result_free": (None, [vp]),
        "ms_score": (c_int, [vp, vp, c_int, pd]),
        "ms_score_ranks": (c_int, [vp, vp, c_int, pi64, c_i32, pd]),
        "ms_dedup_hits": (c_int, [pi64, c_i32, pi32, pi64, pi64, pd, pi8, pu8]),
        "ms_debug_plan_dims": (c_int, [vp, c_int, c_i64, pi32, pi32, pi32, pi32]),
        "ms_debug_plan_rows": (c_int, [vp, pi32, ctypes.POINTER(ctypes.c_int16), pi32, pi32, pi32, pi32, pi32, pi32]),
        "ms_debug_release_scratch": (c_int, []),
        "ms_debug_numa_probe": (c_int, [ctypes.c_char_p, ctypes.c_char_p, pi32, pi32, pi32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    if bool(L.ms_build_flags() & 1) != (LIB_VARIANT == "asm"):
        raise RuntimeError(f"{LIB_PATH}: ms_build_flags() = {L.ms_build_flags()} does not match the variant asked for ({LIB_VARIANT or 'default'}): rebuild (make -C motifscan_amd/csrc)")
    return L


def check(rc):
    """Status code -> the exception class the reference raises for that kind of failure
    (MemoryError / RuntimeError: cscore.c:243-268, 411-430; ValueError: scanner.py:51-54)."""
    if rc == MS_OK:
        return
    msg = lib().ms_last_error().decode("utf-8", "replace")
    if rc == MS_ERR_INVALID:
        raise ValueError(msg)
    if rc == MS_ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def ptr(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def device_count():
    n = ctypes.c_int(0)
    rc = lib().ms_device_count(ctypes.byref(n))
    return n.value if rc == MS_OK else 0


def set_device(device):
    check(lib().ms_set_device(int(device)))


def numa_bind_thread(force=False):
    """Bind the calling thread to the NUMA node of the current device (ms_numa_bind_thread): call it on a rank's main thread before it
    allocates its pinned input buffers.  Returns the node, or -1 if nothing was done (policy off / no NUMA information)."""
    node = ctypes.c_int(-1)
    check(lib().ms_numa_bind_thread(1 if force else 0, ctypes.byref(node)))
    return node.value


def device_name():
    buf = ctypes.create_string_buffer(256)
    check(lib().ms_device_name(buf, 256))
    return buf.value.decode()


# --------------------------------------------------------------------------- handles --

class PwmSet:
    """Device-side PWM set (convert_pwm + get_max_raw_score, cscore.c:36-79)."""

    def __init__(self, values, widths, cutoffs=None):
        self.values = np.ascontiguousarray(values, dtype=np.float64)
        self.widths = np.ascontiguousarray(widths, dtype=np.int32)
        if self.values.size != 4 * int(self.widths.sum()):
            raise ValueError("values must hold 4*width doubles per PWM")
        self.n = len(self.widths)
        cut = None if cutoffs is None else np.ascontiguousarray(cutoffs, dtype=np.float64)
        if cut is not None and cut.size != self.n:
            raise ValueError("need one cutoff per PWM")
        h = ctypes.c_void_p()
        check(lib().ms_pwmset_create(ptr(self.values, ctypes.c_double), ptr(self.widths, ctypes.c_int32),
                                     None if cut is None else ptr(cut, ctypes.c_double), self.n, ctypes.byref(h)))
        self.h = h

    @classmethod
    def from_matrices(cls, matrices, cutoffs=None):
        mats = []
        for m in matrices:
            a = np.asarray(m, dtype=np.float64)
            if a.ndim != 2 or a.shape[0] != 4:
                raise ValueError("each PWM must have exactly 4 rows (A, C, G, T)")
            if a.shape[1] == 0:
                raise ValueError("each PWM needs at least 1 position per row")
            mats.append(np.ascontiguousarray(a))
        widths = np.array([m.shape[1] for m in mats], dtype=np.int32)
        values = np.concatenate([m.ravel() for m in mats]) if mats else np.zeros(0)
        return cls(values, widths, cutoffs)

    def set_cutoffs(self, cutoffs):
        cut = np.ascontiguousarray(cutoffs, dtype=np.float64)
        if cut.size != self.n:
            raise ValueError("need one cutoff per PWM")
        check(lib().ms_pwmset_set_cutoffs(self.h, ptr(cut, ctypes.c_double)))

    def max_raw(self):
        out = np.zeros(self.n, dtype=np.float64)
        check(lib().ms_pwmset_max_raw(self.h, ptr(out, ctypes.c_double)))
        return out

    def plan(self, strand_mask=3, lds_budget=70 * 1024):
        """Host-side view of the pre-filter plan, decoded from the operand image the kernel reads (tests only):
        group_fields [groups][16] motif of the field (-1 empty; both strands: field n = slot n >> 1, even forward, odd reverse;
        one strand: field n = slot n), rows [groups][16][64 columns][4 bases] and bias [groups][16] in units of 1/8,
        group_kb [groups] matrix instructions per row tile, group_cols [groups] columns of the group's fields incl. the bias
        column (16 per instruction; paired rows: 8), group_paired [groups] 0 = plain row, 1 / 2 = field X / Y of a paired row."""
        L = lib()
        nf, ne, nq, nt = (ctypes.c_int32() for _ in range(4))
        check(L.ms_debug_plan_dims(self.h, strand_mask, lds_budget, ctypes.byref(nf), ctypes.byref(ne),
                                   ctypes.byref(nq), ctypes.byref(nt)))
        gf = np.full((nq.value, 16), -1, dtype=np.int32)
        ex = np.zeros(max(ne.value, 1), dtype=np.int32)
        tf = np.zeros(nt.value + 1, dtype=np.int32)
        rows = np.zeros((nq.value, 16, 64, 4), dtype=np.int16)
        bias = np.zeros((nq.value, 16), dtype=np.int32)
        kb = np.zeros(nq.value, dtype=np.int32)
        cols = np.zeros(nq.value, dtype=np.int32)
        paired = np.zeros(nq.value, dtype=np.int32)
        check(L.ms_debug_plan_rows(self.h, ptr(gf, ctypes.c_int32), ptr(rows, ctypes.c_int16), ptr(bias, ctypes.c_int32),
                                   ptr(kb, ctypes.c_int32), ptr(cols, ctypes.c_int32), ptr(paired, ctypes.c_int32),
                                   ptr(ex, ctypes.c_int32), ptr(tf, ctypes.c_int32)))
        return {"n_fast": nf.value, "n_exact": ne.value, "n_tiles": nt.value, "strand_mask": strand_mask, "group_fields": gf,
                "rows": rows, "bias": bias, "group_kb": kb, "group_cols": cols, "group_paired": paired,
                "exact_motifs": ex[:ne.value], "tile_first_group": tf}

    def close(self):
        if getattr(self, "h", None):
            lib().ms_pwmset_free(self.h)
            self.h = None

    __del__ = close


class SeqSet:
    """Device-side packed sequence set (convert_seq, cscore.c:81-114)."""

    def __init__(self, bases, offsets, keep_ascii=False, host_pack_threads=0):
        """host_pack_threads > 0: convert_seq on that many host threads (ms_seqset_create_hostpacked: no kernel is launched) -- the same set."""
        if isinstance(bases, np.ndarray):
            bases = np.ascontiguousarray(bases, dtype=np.uint8)
            buf = bases.ctypes.data_as(ctypes.c_char_p)
            nbytes = bases.size
        else:
            bases = bytes(bases)
            buf = bases
            nbytes = len(bases)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if self.offsets.ndim != 1 or self.offsets.size < 1:
            raise ValueError("offsets must have n_seqs + 1 entries")
        if int(self.offsets[-1]) != nbytes:
            raise ValueError("offsets[-1] must equal the number of bases")
        self.n_seqs = self.offsets.size - 1
        self.n_bases = nbytes
        h = ctypes.c_void_p()
        if host_pack_threads and not keep_ascii:
            check(lib().ms_seqset_create_hostpacked(buf, ptr(self.offsets, ctypes.c_int64), self.n_seqs, int(host_pack_threads), ctypes.byref(h)))
        else:
            check(lib().ms_seqset_create(buf, ptr(self.offsets, ctypes.c_int64), self.n_seqs, int(bool(keep_ascii)),
                                         ctypes.byref(h)))
        self.h = h

    @classmethod
    def from_strings(cls, seqs, keep_ascii=False):
        seqs = seqs if isinstance(seqs, (list, tuple)) else list(seqs)
        offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
        try:
            lens = np.fromiter(map(len, seqs), dtype=np.int64, count=len(seqs))
            if seqs:
                np.cumsum(lens, out=offsets[1:])
            if seqs and lens[0] > 0 and (lens == lens[0]).all():
                # regions of ONE length (the reference's fixed window around the summit, scanner.py:70-79): numpy encodes the list into one
                # fixed-width byte matrix in a single C loop -- 2.6x faster than join + encode for 100 000 x 1 kb
                raw = np.array(seqs, dtype="S%d" % int(lens[0])).view(np.uint8).reshape(-1)
                if raw.size != int(offsets[-1]):
                    raise TypeError("not a flat list of strings")
            else:                                 # ONE join + ONE encode (plain-ASCII strings: a base is a byte)
                raw = "".join(seqs).encode("ascii")
        except (TypeError, ValueError, UnicodeEncodeError):   # bytes among them, or non-ASCII text (every such byte scores as 'N', cscore.c:92-111)
            bs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in seqs]
            if bs:
                offsets[1:] = np.cumsum([len(b) for b in bs])
            raw = b"".join(bs)
        return cls(raw, offsets, keep_ascii)

    @classmethod
    def from_device(cls, device_ptr, offsets):
        """ASCII already resident in device memory (e.g. a torch uint8 tensor's data_ptr())."""
        self = cls.__new__(cls)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.n_seqs = self.offsets.size - 1
        self.n_bases = int(self.offsets[-1])
        h = ctypes.c_void_p()
        check(lib().ms_seqset_from_device(ctypes.c_void_p(int(device_ptr)), ptr(self.offsets, ctypes.c_int64),
                                          self.n_seqs, ctypes.byref(h)))
        self.h = h
        return self

    def repack(self):
        check(lib().ms_seqset_repack(self.h))

    def close(self):
        if getattr(self, "h", None):
            lib().ms_seqset_free(self.h)
            self.h = None

    __del__ = close


class ResidentGenome:
    """A genome packed once into HBM; regions are cut out on the device (no per-region host fetch).

    Quacks like the part of `motifscan.genome.Genome` the Scanner reads: `chrom_sizes` and
    `fetch_sequence(chrom, start, end)` (served from a host copy only if keep_host=True)."""

    def __init__(self, chroms, keep_host=False):
        """chroms: dict name -> str/bytes/uint8 array (insertion order = chromosome index)."""
        self.names = list(chroms.keys())
        self.index = {n: i for i, n in enumerate(self.names)}
        raws = [v.encode() if isinstance(v, str) else (v.tobytes() if isinstance(v, np.ndarray) else bytes(v))
                for v in chroms.values()]
        self.chrom_sizes = {n: len(r) for n, r in zip(self.names, raws)}
        self.offsets = np.zeros(len(raws) + 1, dtype=np.int64)
        if raws:
            self.offsets[1:] = np.cumsum([len(r) for r in raws])
        self._host = dict(zip(self.names, raws)) if keep_host else None
        h = ctypes.c_void_p()
        check(lib().ms_genome_create(b"".join(raws), ptr(self.offsets, ctypes.c_int64), len(raws), ctypes.byref(h)))
        self.h = h

    # ---- FASTA -> packed genome file -> resident genome (motifscan_amd/genome.py has the file format) ----
    @classmethod
    def from_packed(cls, packed):
        """The planes of a genome.PackedGenome (arrays or a memory-mapped genome file) uploaded as they are: 0.375 B per base over the
        link, no ASCII, no pack kernel (ms_genome_create_packed).  fetch_sequence is served by the packed host side."""
        g = cls.__new__(cls)
        g.names = list(packed.names)
        g.index = dict(packed.index)
        g.chrom_sizes = dict(packed.chrom_sizes)
        g.offsets = np.ascontiguousarray(packed.offsets, dtype=np.int64)
        g._host = None
        g._packed = packed
        codes = np.ascontiguousarray(packed.codes, dtype=np.uint32)          # (a memory map: read sequentially here, once)
        nmask = np.ascontiguousarray(packed.nmask, dtype=np.uint32)
        h = ctypes.c_void_p()
        check(lib().ms_genome_create_packed(ptr(codes, ctypes.c_uint32), ptr(nmask, ctypes.c_uint32), ptr(g.offsets, ctypes.c_int64),
                                            len(g.names), ctypes.byref(h)))
        g.h = h
        return g

    @classmethod
    def from_fasta(cls, path, n_threads=None):
        """Read a FASTA (plain reader, optional .fai check), pack it on host threads, upload the planes.  To pay the read + pack once
        per genome instead of once per process: ResidentGenome.from_fasta(fa).save(path) -- then ResidentGenome.load(path)."""
        from . import genome as _genome
        return cls.from_packed(_genome.PackedGenome.from_fasta(path, n_threads=n_threads))

    @classmethod
    def load(cls, path, verify=True):
        from . import genome as _genome
        return cls.from_packed(_genome.PackedGenome.load(path, verify=verify))

    def packed(self):
        """The genome as a genome.PackedGenome: the one it was made from, or -- for a genome packed on the device from strings -- its planes
        copied back (ms_genome_packed_host; case and IUPAC letters come from the host copy if one was kept, else they are lost: N / upper case)."""
        if getattr(self, "_packed", None) is not None:
            return self._packed
        from . import genome as _genome
        if self._host is not None:
            return _genome.PackedGenome.from_arrays(self.names, [self._host[n] for n in self.names])
        n = int(self.offsets[-1])
        units = (n + 31) // 32
        codes, nmask = np.zeros(2 * units, dtype=np.uint32), np.zeros(units, dtype=np.uint32)
        check(lib().ms_genome_packed_host(self.h, ptr(codes, ctypes.c_uint32), ptr(nmask, ctypes.c_uint32)))
        return _genome.PackedGenome(self.names, self.offsets, codes, nmask)

    def save(self, path):
        return self.packed().save(path)

    @property
    def chroms(self):
        return sorted(self.names)                        # genome/__init__.py:88-99

    def fetch_sequence(self, chrom, start, end):
        if self._host is not None:
            return self._host[chrom][start:end].decode()
        if getattr(self, "_packed", None) is not None:
            return self._packed.fetch_sequence(chrom, start, end)
        raise RuntimeError("ResidentGenome was created without keep_host: sequences live on the device only")

    def extract(self, chrom_idx, starts, ends):
        """SeqSet of the regions (chromosome indices, 0-based half-open, already clipped)."""
        ci = np.ascontiguousarray(chrom_idx, dtype=np.int32)
        st = np.ascontiguousarray(starts, dtype=np.int64)
        en = np.ascontiguousarray(ends, dtype=np.int64)
        if not (len(ci) == len(st) == len(en)):
            raise ValueError("chrom / start / end must have the same length")
        sq = SeqSet.__new__(SeqSet)
        h = ctypes.c_void_p()
        check(lib().ms_seqset_from_genome(self.h, ptr(ci, ctypes.c_int32), ptr(st, ctypes.c_int64), ptr(en, ctypes.c_int64),
                                          len(ci), ctypes.byref(h)))
        sq.h = h
        sq.n_seqs = len(ci)
        sq.offsets = None                         # lives on the device; not needed on the host
        nb = ctypes.c_int64()
        check(lib().ms_seqset_size(sq.h, None, ctypes.byref(nb)))
        sq.n_bases = nb.value
        return sq

    def close(self):
        if getattr(self, "h", None):
            lib().ms_genome_free(self.h)
            self.h = None

    __del__ = close


def _owned_array(address, ctype, n, dtype, owner):
    """A numpy array over n items of library-owned memory at `address` that keeps `owner` (the ScanResult / pinned block that
    frees the memory) alive for as long as ANY array derived from it lives.  numpy collapses `.base` to the object that owns
    the memory -- np.asarray(view), slices, reshapes all end up referencing the ctypes array below, not an ndarray subclass in
    between -- so the owner hangs on that object (an owner kept on an ndarray subclass was lost by np.asarray)."""
    raw = (ctype * max(int(n), 1)).from_address(int(address))
    raw._owner = owner
    return np.frombuffer(raw, dtype=dtype, count=int(n))


class ScanResult:
    """Hits of one ms_scan call, in the reference's order (cscore.c:336-390, 443-471)."""

    def __init__(self, handle, n_pwms):
        self.h = handle
        self.n_pwms = n_pwms
        n = ctypes.c_int64()
        check(lib().ms_result_num_hits(self.h, ctypes.byref(n)))
        self.n_hits = n.value
        self.motif_offsets = np.zeros(n_pwms + 1, dtype=np.int64)
        check(lib().ms_result_motif_offsets(self.h, ptr(self.motif_offsets, ctypes.c_int64)))
        self._hits = None
        self._hits_owned = True

    def hits(self, copy=True, packed=False, motif=True):
        """dict of numpy arrays: seq_idx, pos, score, strand (1 '+', 2 '-'), motif (motif=False leaves the per-hit motif
        index out: it is motif_offsets expanded, 4 bytes per hit that few callers read).
        copy=False returns views of the library's pinned host buffers; the views keep this result alive (they are only
        invalidated by an explicit close() or dedup()).  packed=True moves 16 instead of 25 bytes per hit over the host
        link (ms_result_hits_packed_host) and unpacks on the host -- the arrays are then always fresh copies."""
        if packed:
            return self._hits_packed(None if packed is True else int(packed))
        if self._hits is None or (copy and not self._hits_owned):
            n = self.n_hits
            ps, pp = ctypes.POINTER(ctypes.c_int64)(), ctypes.POINTER(ctypes.c_int64)()
            pv, pd_ = ctypes.POINTER(ctypes.c_double)(), ctypes.POINTER(ctypes.c_int8)()
            check(lib().ms_result_hits_host(self.h, ctypes.byref(ps), ctypes.byref(pp), ctypes.byref(pv), ctypes.byref(pd_)))
            if n:
                spec = ((ps, ctypes.c_int64, np.int64), (pp, ctypes.c_int64, np.int64), (pv, ctypes.c_double, np.float64), (pd_, ctypes.c_int8, np.int8))
                arrs = [_owned_array(ctypes.cast(q, ctypes.c_void_p).value, ct, n, dt, self) for q, ct, dt in spec]
                if copy:
                    arrs = [a.copy() for a in arrs]
            else:
                arrs = [np.zeros(0, dtype=t) for t in (np.int64, np.int64, np.float64, np.int8)]
            self._hits = {"seq_idx": arrs[0], "pos": arrs[1], "score": arrs[2], "strand": arrs[3],
                          "motif_offsets": self.motif_offsets}
            self._hits_owned = copy or n == 0
        out = self._hits
        if motif and "motif" not in out:
            out["motif"] = np.repeat(np.arange(self.n_pwms, dtype=np.int32), np.diff(self.motif_offsets))
        if not self._hits_owned:
            self._hits = None                            # the views reference this object: caching them here would be a cycle
        return out

    def packed_form(self):
        """12, 16 or 0: the compact form a batch stream's copy-out left in this result (ms_result_packed_form)."""
        b = ctypes.c_int32()
        check(lib().ms_result_packed_form(self.h, ctypes.byref(b)))
        return b.value

    def _hits_packed(self, form=None):
        """form: None = whatever the result holds already (a stream's copy-out), else 12 / 16."""
        n = self.n_hits
        if form is None:
            form = self.packed_form() or 16
        if form == 12:
            pc, pv, sh = ctypes.POINTER(ctypes.c_uint32)(), ctypes.POINTER(ctypes.c_double)(), ctypes.c_int32()
            check(lib().ms_result_hits_packed12_host(self.h, ctypes.byref(pc), ctypes.byref(pv), ctypes.byref(sh)))
            if n:
                coord = np.ctypeslib.as_array(pc, shape=(n,))
                score_ = np.ctypeslib.as_array(pv, shape=(n,)).copy()
                seq_idx = (coord >> np.uint32(sh.value)).astype(np.int64)
                pos = ((coord & np.uint32((1 << sh.value) - 1)) >> np.uint32(1)).astype(np.int64)
                strand = ((coord & np.uint32(1)) + np.uint32(1)).astype(np.int8)
            else:
                seq_idx, pos, score_, strand = (np.zeros(0, dtype=t) for t in (np.int64, np.int64, np.float64, np.int8))
            motif = np.repeat(np.arange(self.n_pwms, dtype=np.int32), np.diff(self.motif_offsets))
            return {"seq_idx": seq_idx, "pos": pos, "score": score_, "strand": strand, "motif": motif, "motif_offsets": self.motif_offsets}
        pc, pv = ctypes.POINTER(ctypes.c_uint64)(), ctypes.POINTER(ctypes.c_double)()
        check(lib().ms_result_hits_packed_host(self.h, ctypes.byref(pc), ctypes.byref(pv)))
        if n:
            coord = np.ctypeslib.as_array(pc, shape=(n,))
            score_ = np.ctypeslib.as_array(pv, shape=(n,)).copy()
            seq_idx = (coord >> np.uint64(32)).astype(np.int64)
            pos = ((coord & np.uint64(0xFFFFFFFF)) >> np.uint64(1)).astype(np.int64)
            strand = ((coord & np.uint64(1)) + np.uint64(1)).astype(np.int8)
        else:
            seq_idx, pos, score_, strand = (np.zeros(0, dtype=t) for t in (np.int64, np.int64, np.float64, np.int8))
        motif = np.repeat(np.arange(self.n_pwms, dtype=np.int32), np.diff(self.motif_offsets))
        return {"seq_idx": seq_idx, "pos": pos, "score": score_, "strand": strand, "motif": motif,
                "motif_offsets": self.motif_offsets}

    def dedup(self, pwms):
        """Device-side de-duplication in place (scanner.py:156-193)."""
        check(lib().ms_result_dedup(self.h, pwms.h))
        n = ctypes.c_int64()
        check(lib().ms_result_num_hits(self.h, ctypes.byref(n)))
        self.n_hits = n.value
        check(lib().ms_result_motif_offsets(self.h, ptr(self.motif_offsets, ctypes.c_int64)))
        self._hits = None
        return self

    def site_tables(self, n_seqs):
        """(n_sites int32 [P][R], max_score float64 [P][R], NaN = no site)."""
        n_sites = np.zeros((self.n_pwms, n_seqs), dtype=np.int32)
        max_score = np.full((self.n_pwms, n_seqs), np.nan, dtype=np.float64)
        check(lib().ms_result_site_tables(self.h, ptr(n_sites, ctypes.c_int32), ptr(max_score, ctypes.c_double)))
        return n_sites, max_score

    def region_counts(self):
        out = np.zeros(self.n_pwms, dtype=np.int64)
        check(lib().ms_result_region_counts(self.h, ptr(out, ctypes.c_int64)))
        return out

    def region_counts_device_ptr(self):
        p = ctypes.c_void_p()
        check(lib().ms_result_region_counts_device(self.h, ctypes.byref(p)))
        return p.value

    def stats(self):
        s = ScanStats()
        check(lib().ms_result_stats(self.h, ctypes.byref(s)))
        return s.as_dict()

    def close(self):
        if getattr(self, "h", None):
            lib().ms_result_free(self.h)
            self.h = None

    __del__ = close


def host_pack(bases, offsets):
    """The host packer alone (ms_debug_host_pack; no device): (codes uint32 [2 x units], nmask uint32 [units], blk2reg int32 [blocks],
    blkinfo int32 [blocks][4]) of a sequence set -- the layout the device's pack / hint kernels write."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = int(offsets[-1])
    units, blocks = (n + 31) // 32, (n + 63) // 64 + 1
    codes, nmask = np.zeros(2 * units, dtype=np.uint32), np.zeros(units, dtype=np.uint32)
    blk, info = np.zeros(blocks, dtype=np.int32), np.zeros((blocks, 4), dtype=np.int32)
    check(lib().ms_debug_host_pack(bases.ctypes.data_as(ctypes.c_char_p), ptr(offsets, ctypes.c_int64), offsets.size - 1,
                                   ptr(codes, ctypes.c_uint32), ptr(nmask, ctypes.c_uint32), ptr(blk, ctypes.c_int32), ptr(info, ctypes.c_int32)))
    return codes, nmask, blk, info


def host_pool_stats():
    """The pinned-block cache (ms_host_pool_stats): dict(hits, misses, driver_frees, ms_in_driver)."""
    out = (ctypes.c_uint64 * 4)()
    check(lib().ms_host_pool_stats(out))
    return {"hits": int(out[0]), "misses": int(out[1]), "driver_frees": int(out[2]), "ms_in_driver": out[3] / 1e6}


def pool_stats():
    """HBM block cache of the calling thread's device (ms_device_pool_stats)."""
    out = (ctypes.c_uint64 * 6)()
    check(lib().ms_device_pool_stats(out))
    return {"hits": int(out[0]), "misses": int(out[1]), "driver_frees": int(out[2]), "driver_ms": out[3] / 1e6,
            "cached_bytes": int(out[4]), "cached_blocks": int(out[5])}


_warned_exact_only = set()


def _fence_exact_only(res, pwms, flags=0):
    """motifs the pre-filter cannot take -- wider than 63 columns, max_raw <= 0, non-finite entries, a cutoff under the
    quantiser's floor -- are scored in fp64 at EVERY window (exact_tiled_kernel: correct, the reference's arithmetic, the motif's table in LDS --
    and still ~400 x the cost per motif of the matrix-core path; the reference itself has no width limit, cscore.c:50-51).  Never silent:
    one warning per PWM set with the number of such motifs (MS_SCAN_EXACT_ONLY scans -- validation -- are the caller's own choice)."""
    if (flags & MS_SCAN_EXACT_ONLY) or id(pwms) in _warned_exact_only:
        return res
    st = res.stats()
    if st["n_pwms_exact"] > 0 and st["n_pwms_exact"] * st["n_bases"] >= 10_000_000:       # (a handful of short sequences costs nothing either way)
        _warned_exact_only.add(id(pwms))
        import warnings
        warnings.warn(f"{st['n_pwms_exact']} of {st['n_pwms']} PWMs cannot take the matrix-core pre-filter (wider than 63 columns, max_raw <= 0, "
                      f"non-finite entries or a cutoff below the quantiser's floor) and are scored in fp64 at every window: same results, "
                      f"roughly 400 x the device time per such motif (fp64 stage of this scan: {st['ms_exact']:.1f} ms of {st['ms_total']:.1f} ms)", RuntimeWarning, stacklevel=3)
    return res


def scan(pwms, seqs, strand_mask=3, flags=MS_SCAN_DEFAULT):
    h = ctypes.c_void_p()
    check(lib().ms_scan(pwms.h, seqs.h, int(strand_mask), int(flags), ctypes.byref(h)))
    return _fence_exact_only(ScanResult(h, pwms.n), pwms, int(flags))


def scan_sweep(pwms, genome, chrom, begin, end, window, stride, strand_mask=3, flags=MS_SCAN_DEFAULT):
    """Scan the windows [begin + k*stride, begin + k*stride + window) of one chromosome of a ResidentGenome.  Same
    result as scan() over those windows as separate regions (seq_idx = k), with every base scored once."""
    ci = genome.index[chrom] if isinstance(chrom, str) else int(chrom)
    h = ctypes.c_void_p()
    check(lib().ms_scan_sweep(pwms.h, genome.h, ci, int(begin), int(end), int(window), int(stride), int(strand_mask),
                              int(flags), ctypes.byref(h)))
    return _fence_exact_only(ScanResult(h, pwms.n), pwms, int(flags))


class PinnedBuffer:
    """Page-locked host memory (ms_host_alloc) as a uint8 numpy array: uploads from it run at the full link rate and
    overlap with scans."""

    def __init__(self, nbytes):
        p = ctypes.c_void_p()
        check(lib().ms_host_alloc(int(nbytes), ctypes.byref(p)))
        self.ptr = p
        self.nbytes = int(nbytes)

    @property
    def array(self):
        """uint8 view of the block; every view (and anything numpy derives from it) keeps the block alive.  Made on demand: a
        view stored on the object itself would be a reference cycle and delay the release of the pinned memory to the cycle GC."""
        if not getattr(self, "ptr", None):
            raise ValueError("the pinned buffer is closed")
        return _owned_array(self.ptr.value, ctypes.c_uint8, self.nbytes, np.uint8, self)

    def close(self):
        if getattr(self, "ptr", None):
            lib().ms_host_free(self.ptr)
            self.ptr = None

    __del__ = close


class Stream:
    """ms_stream: batches of regions (or spans of a window sweep) flow through upload + pack | scan | copy-out, the three
    stages of consecutive batches overlapped on one device.  Results come back in submission order."""

    def __init__(self, pwms, strand_mask=3, flags=0, depth=2):
        self.pwms = pwms                               # keep alive
        self.flags = int(flags)
        h = ctypes.c_void_p()
        check(lib().ms_stream_create(pwms.h, int(strand_mask), int(flags), int(depth), ctypes.byref(h)))
        self.h = h
        n = ctypes.c_int()
        check(lib().ms_stream_capacity(self.h, ctypes.byref(n)))
        self.capacity = n.value
        self._keep = []                                # buffers borrowed by batches still in flight

    @property
    def in_flight(self):
        n = ctypes.c_int()
        check(lib().ms_stream_in_flight(self.h, ctypes.byref(n)))
        return n.value

    def submit(self, bases, offsets, counts_only=False):
        bases = np.ascontiguousarray(bases, dtype=np.uint8) if isinstance(bases, np.ndarray) else np.frombuffer(bytes(bases), dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if offsets.ndim != 1 or offsets.size < 1 or int(offsets[0]) != 0 or int(offsets[-1]) != bases.size:
            raise ValueError("offsets must run from 0 to the number of bases")
        fn = lib().ms_stream_submit_counts_only if counts_only else lib().ms_stream_submit
        check(fn(self.h, ctypes.c_void_p(bases.ctypes.data), ptr(offsets, ctypes.c_int64), offsets.size - 1))
        self._keep.append(bases)

    def submit_regions(self, genome, chrom_idx, starts, ends):
        """A batch of regions of a ResidentGenome (chromosome indices, 0-based half-open): cut on the device while the previous
        batch is scanned."""
        ci = np.ascontiguousarray(chrom_idx, dtype=np.int32)
        st = np.ascontiguousarray(starts, dtype=np.int64)
        en = np.ascontiguousarray(ends, dtype=np.int64)
        if not (ci.size == st.size == en.size):
            raise ValueError("chrom_idx, starts and ends must have one entry per region")
        check(lib().ms_stream_submit_regions(self.h, genome.h, ptr(ci, ctypes.c_int32), ptr(st, ctypes.c_int64), ptr(en, ctypes.c_int64), ci.size))
        self._keep.append(genome)

    def submit_span(self, bases, window, stride):
        bases = np.ascontiguousarray(bases, dtype=np.uint8) if isinstance(bases, np.ndarray) else np.frombuffer(bytes(bases), dtype=np.uint8)
        check(lib().ms_stream_submit_span(self.h, ctypes.c_void_p(bases.ctypes.data), bases.size, int(window), int(stride)))
        self._keep.append(bases)

    def next(self):
        """The oldest batch's ScanResult (hits already on the host), or None when nothing is in flight."""
        if self.in_flight == 0:
            self._keep.clear()
            return None
        h = ctypes.c_void_p()
        try:
            check(lib().ms_stream_next(self.h, ctypes.byref(h)))
        finally:
            if self._keep:
                self._keep.pop(0)                      # returned or failed: either way the batch no longer borrows its buffer
        return _fence_exact_only(ScanResult(h, self.pwms.n), self.pwms, MS_SCAN_EXACT_ONLY if self.flags & MS_STREAM_EXACT_ONLY else 0) if h.value else None

    def stats(self):
        """ms_stream_stats: per stage {batches, ms_work, ms_wait_in, ms_wait_out}; the stage that waits least bounds the stream."""
        out = (ctypes.c_double * 12)()
        check(lib().ms_stream_stats(self.h, out))
        return {name: {"batches": int(out[4 * k]), "ms_work": out[4 * k + 1], "ms_wait_in": out[4 * k + 2], "ms_wait_out": out[4 * k + 3]}
                for k, name in enumerate(("upload", "scan", "copy_out"))}

    def close(self):
        if getattr(self, "h", None):
            lib().ms_stream_free(self.h)
            self.h = None
            self._keep = []

    __del__ = close


def merge_hits(parts, n_pwms):
    """Concatenate per-batch hit dicts into the single-call result: parts = [(hits dict, seq_idx offset), ...] in batch
    order.  Within a motif the reference orders hits by sequence index (cscore.c:336-390), so motif p's hits are batch 0's
    hits of p, then batch 1's, ... -- one slice copy per (motif, batch)."""
    offs = np.zeros(n_pwms + 1, dtype=np.int64)
    for h, _ in parts:
        offs[1:] += np.diff(h["motif_offsets"])
    offs = np.concatenate([[0], np.cumsum(offs[1:])]).astype(np.int64)
    n = int(offs[-1])
    out = {"seq_idx": np.empty(n, np.int64), "pos": np.empty(n, np.int64), "score": np.empty(n, np.float64),
           "strand": np.empty(n, np.int8)}
    cur = offs[:-1].copy()
    for h, shift in parts:
        mo = h["motif_offsets"]
        for p in np.nonzero(np.diff(mo))[0].tolist():
            a, b = int(mo[p]), int(mo[p + 1])
            d = int(cur[p])
            out["seq_idx"][d:d + b - a] = h["seq_idx"][a:b] + shift
            out["pos"][d:d + b - a] = h["pos"][a:b]
            out["score"][d:d + b - a] = h["score"][a:b]
            out["strand"][d:d + b - a] = h["strand"][a:b]
            cur[p] += b - a
    out["motif"] = np.repeat(np.arange(n_pwms, dtype=np.int32), np.diff(offs))
    out["motif_offsets"] = offs
    return out


def scan_stream(pwms, batches, strand_mask=3, flags=0, depth=2, packed=False, stage_stats=None, host_pack=False):
    """Generator: push (bases, offsets) batches -- or (ResidentGenome, chrom_idx, starts, ends) batches of a genome that sits in
    HBM -- through a Stream, yield each batch's ScanResult in order (the caller closes them).  Keeps the stream as full as its capacity allows.  stage_stats: a dict that receives Stream.stats() at the end.
    A (bases, offsets, True) batch is COUNTS ONLY (ms_stream_submit_counts_only): its hits stay on the device, only the per-motif
    region counts are read -- what the reference does with the control regions (cli/scan.py:81-89 -> stats.py:29-31).
    host_pack: the upload stage packs the batches on host threads (MS_STREAM_HOST_PACK: no kernel beside the scan).
    packed: True / 16 = the 16-byte compact copy-out, 12 = the 12-byte form for every batch that fits it (MS_STREAM_PACKED12)."""
    st = Stream(pwms, strand_mask, flags | (MS_STREAM_PACKED12 if packed == 12 else (MS_STREAM_PACKED if packed else 0)) | (MS_STREAM_HOST_PACK if host_pack else 0), depth)
    try:
        for batch in batches:                             # (bases, offsets), or (ResidentGenome, chrom_idx, starts, ends)
            while st.in_flight >= st.capacity:
                yield st.next()
            if len(batch) == 4:
                st.submit_regions(*batch)
            else:
                st.submit(*batch)                         # (bases, offsets) or (bases, offsets, counts_only)
        while st.in_flight:
            yield st.next()
        if stage_stats is not None:
            stage_stats.update(st.stats())
    finally:
        st.close()


def sweep_stream(pwms, chroms, window, stride, max_span_bases, strand_mask=3, flags=0, depth=2, spans=None, packed=False,
                 stage_stats=None):
    """Host-streamed window sweep over a whole genome (BASELINE configs[4]): chroms = list of uint8 arrays (host memory,
    ideally pinned); the windows [k*stride, k*stride + window) of every chromosome are cut into spans of at most
    max_span_bases bases (ms_sweep_spans), and the spans flow through a Stream (upload + pack | scan-once + hand-out |
    copy-out).  Yields (span, ScanResult) in order; span = (chrom, begin, end, first_window, n_windows) and the result's
    seq_idx counts windows from the span's first one: add span[3] for the genome-wide window index.  `spans` restricts
    the sweep to a subset (a rank's share)."""
    if spans is None:
        spans = sweep_spans([len(c) for c in chroms], window, stride, max_span_bases)
    st = Stream(pwms, strand_mask, flags | (MS_STREAM_PACKED12 if packed == 12 else (MS_STREAM_PACKED if packed else 0)), depth)
    try:
        pending = []
        for sp in spans:
            while st.in_flight >= st.capacity:
                yield pending.pop(0), st.next()
            st.submit_span(chroms[sp[0]][sp[1]:sp[2]], window, stride)
            pending.append(sp)
        while st.in_flight:
            yield pending.pop(0), st.next()
        if stage_stats is not None:
            stage_stats.update(st.stats())
    finally:
        st.close()


def sweep_spans(chrom_lens, window, stride, max_span_bases):
    """ms_sweep_spans: [(chrom, begin, end, first_window, n_windows), ...] for the windows [k*stride, k*stride + window)
    of every chromosome, in spans of at most max_span_bases bases."""
    lens = np.ascontiguousarray(chrom_lens, dtype=np.int64)
    n = ctypes.c_int64()
    check(lib().ms_sweep_spans(ptr(lens, ctypes.c_int64), len(lens), int(window), int(stride), int(max_span_bases), None, 0,
                               ctypes.byref(n)))
    arr = (Span * max(n.value, 1))()
    check(lib().ms_sweep_spans(ptr(lens, ctypes.c_int64), len(lens), int(window), int(stride), int(max_span_bases), arr, n.value,
                               ctypes.byref(n)))
    return [(a.chrom, a.begin, a.end, a.first_window, a.n_windows) for a in arr[:n.value]]


def scan_regions_once(pwms, genome, chrom_idx, starts, ends, strand_mask=3, flags=MS_SCAN_DEFAULT):
    """ms_scan_regions_once: the regions (chromosome indices, 0-based half-open, clipped; any order, any overlap) of a
    ResidentGenome, with the result of extract() + scan() over them but every base of their union scored once."""
    ci = np.ascontiguousarray(chrom_idx, dtype=np.int32)
    st = np.ascontiguousarray(starts, dtype=np.int64)
    en = np.ascontiguousarray(ends, dtype=np.int64)
    if not (len(ci) == len(st) == len(en)):
        raise ValueError("chrom / start / end must have the same length")
    h = ctypes.c_void_p()
    check(lib().ms_scan_regions_once(pwms.h, genome.h, ptr(ci, ctypes.c_int32), ptr(st, ctypes.c_int64), ptr(en, ctypes.c_int64),
                                     len(ci), int(strand_mask), int(flags), ctypes.byref(h)))
    return _fence_exact_only(ScanResult(h, pwms.n), pwms, int(flags))


def union_bases(chrom_idx, starts, ends):
    """Bases covered by at least one region (the regions' union), by chromosome-wise interval merging."""
    ci = np.asarray(chrom_idx, dtype=np.int64)
    st = np.asarray(starts, dtype=np.int64)
    en = np.asarray(ends, dtype=np.int64)
    if len(ci) == 0:
        return 0
    o = np.lexsort((st, ci))
    c, s, e = ci[o], st[o], en[o]
    big = int(e.max()) + 1
    reach = np.maximum.accumulate(e + c * big) - c * big          # furthest end seen so far on this chromosome
    prev = np.concatenate([[0], reach[:-1]])
    prev[np.concatenate([[True], c[1:] != c[:-1]])] = 0
    return int(np.maximum(e - np.maximum(s, prev), 0).sum())


def score(pwms, seqs, strand_mask=3):
    out = np.zeros((pwms.n, seqs.n_seqs), dtype=np.float64)
    check(lib().ms_score(pwms.h, seqs.h, int(strand_mask), ptr(out, ctypes.c_double)))
    return out


def score_ranks(pwms, seqs, ranks, strand_mask=3):
    """out[p][k] = the score at 0-based rank ranks[k] of PWM p's scores over all sequences, descending."""
    ranks = np.ascontiguousarray(ranks, dtype=np.int64)
    out = np.zeros((pwms.n, len(ranks)), dtype=np.float64)
    check(lib().ms_score_ranks(pwms.h, seqs.h, int(strand_mask), ptr(ranks, ctypes.c_int64), len(ranks),
                               ptr(out, ctypes.c_double)))
    return out


def release_scratch():
    check(lib().ms_debug_release_scratch())


def dedup_keep(motif_offsets, widths, seq_idx, pos, score_, strand):
    motif_offsets = np.ascontiguousarray(motif_offsets, dtype=np.int64)
    widths = np.ascontiguousarray(widths, dtype=np.int32)
    seq_idx = np.ascontiguousarray(seq_idx, dtype=np.int64)
    pos = np.ascontiguousarray(pos, dtype=np.int64)
    score_ = np.ascontiguousarray(score_, dtype=np.float64)
    strand = np.ascontiguousarray(strand, dtype=np.int8)
    keep = np.ones(len(seq_idx), dtype=np.uint8)
    check(lib().ms_dedup_hits(ptr(motif_offsets, ctypes.c_int64), len(widths), ptr(widths, ctypes.c_int32),
                              ptr(seq_idx, ctypes.c_int64), ptr(pos, ctypes.c_int64), ptr(score_, ctypes.c_double),
                              ptr(strand, ctypes.c_int8), ptr(keep, ctypes.c_uint8)))
    return keep.astype(bool)
