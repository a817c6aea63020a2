#!/usr/bin/env python3
"""N3 in numbers: FASTA -> packed genome file -> genome resident in HBM, against packing the same chromosomes from strings on the device.
python tools/genome_file_time.py [Mbp]   (GPU box; writes under /tmp)"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motifscan_amd import _lib, genome, synth
mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 400
_lib.set_device(0)
n_chr = 8
bases, off = synth.make_regions(n_chr, mbp * 1_000_000 // n_chr, seed=5, frac_n=0.02)
d = tempfile.mkdtemp(prefix="msg_")
fa = os.path.join(d, "g.fa")
t = time.perf_counter()
with open(fa, "wb") as fh:
    for k in range(n_chr):
        s = bases[int(off[k]):int(off[k + 1])]
        fh.write(f">chr{k + 1} synthetic\n".encode())
        n_full, rest = divmod(s.size, 60)
        body = np.full((n_full, 61), 10, dtype=np.uint8)
        body[:, :60] = s[:n_full * 60].reshape(n_full, 60)
        fh.write(body.tobytes())
        if rest:
            fh.write(s[n_full * 60:].tobytes() + b"\n")
print(f"wrote {os.path.getsize(fa) / 1e6:.0f} MB FASTA in {time.perf_counter() - t:.1f} s", flush=True)
t = time.perf_counter(); names, seqs = genome.read_fasta(fa); t_read = time.perf_counter() - t
assert [s.size for s in seqs] == np.diff(off).tolist() and all(np.array_equal(s, bases[int(off[k]):int(off[k + 1])]) for k, s in enumerate(seqs))
t = time.perf_counter(); pg = genome.PackedGenome.from_arrays(names, seqs); t_pack = time.perf_counter() - t
path = os.path.join(d, "g.msg")
t = time.perf_counter(); pg.save(path); t_save = time.perf_counter() - t
t = time.perf_counter(); pg2 = genome.PackedGenome.load(path, verify=True); t_load = time.perf_counter() - t
t = time.perf_counter(); pg3 = genome.PackedGenome.load(path, verify=False); rg = _lib.ResidentGenome.from_packed(pg3); t_up = time.perf_counter() - t
t = time.perf_counter(); rd = _lib.ResidentGenome({f"chr{k + 1}": bases[int(off[k]):int(off[k + 1])] for k in range(n_chr)}); t_dev = time.perf_counter() - t
pa, pb = rd.packed(), rg.packed()
same = np.array_equal(np.asarray(pa.codes), np.asarray(pb.codes)) and np.array_equal(np.asarray(pa.nmask), np.asarray(pb.nmask))
a = int(off[3]) + 12345
assert rg.fetch_sequence("chr4", 12345, 12345 + 300) == bases[a:a + 300].tobytes().decode()
print(f"{mbp} Mbp, {n_chr} chromosomes: read_fasta {t_read:.2f} s ({os.path.getsize(fa) / t_read / 1e9:.2f} GB/s), host pack + case plane {t_pack:.2f} s, save {t_save:.2f} s "
      f"({os.path.getsize(path) / 1e6:.0f} MB file = {os.path.getsize(path) / (mbp * 1e6):.3f} B/base), load + CRC {t_load:.2f} s, map + upload (ms_genome_create_packed) {t_up:.2f} s; "
      f"from strings on the device (ASCII H2D + pack_kernel) {t_dev:.2f} s; planes identical: {same}")
