"""
GPU tests on the BASELINE.json workloads themselves (configs[3] shard, configs[4] shard), on the batch-stream / host-streamed
sweep API, and on the rows either side of the path fed from DEVICE output (N4 writers, make_motif_sites).  Run with -m gpu.
Everything goes through the C-ABI; the oracle and the goldens are only the checker.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from motifscan_amd import _lib, cscore, dist, formats, scanner, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def device():
    if _lib.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests need an MI355X (there is no CPU fallback)")
    _lib.set_device(0)


def assert_same_hits(got, want):
    assert np.array_equal(got["motif_offsets"], want["motif_offsets"])
    assert np.array_equal(got["seq_idx"], want["seq_idx"])
    assert np.array_equal(got["pos"], want["pos"])
    assert np.array_equal(np.asarray(got["strand"]).astype(np.int32), np.asarray(want["strand"]).astype(np.int32))
    assert np.array_equal(got["score"], want["score"])          # bit-exact fp64


def order_key_increasing(h, pos_bits, chunk=1 << 26):
    """(motif, seq, pos, strand) strictly increasing = the reference's order (cscore.c:336-390), checked in chunks."""
    n = len(h["pos"])
    prev = None
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        key = (h["motif"][a:b].astype(np.int64) << 52) | (h["seq_idx"][a:b] << (pos_bits + 1)) | (h["pos"][a:b] << 1) | (h["strand"][a:b] == 2)
        if prev is not None and not key[0] > prev:
            return False
        if not (np.diff(key) > 0).all():
            return False
        prev = key[-1]
    return True


def recount_regions(h, n_pwms):
    """per motif: number of distinct sequences with a hit (stats.py:29-31) -- recomputed from the hit list itself."""
    if len(h["pos"]) == 0:
        return np.zeros(n_pwms, dtype=np.int64)
    motif = h["motif"] if "motif" in h else np.repeat(np.arange(n_pwms), np.diff(h["motif_offsets"]))
    new_pair = np.ones(len(h["pos"]), dtype=bool)
    new_pair[1:] = (motif[1:] != motif[:-1]) | (h["seq_idx"][1:] != h["seq_idx"][:-1])
    return np.bincount(motif[new_pair], minlength=n_pwms).astype(np.int64)


# ------------------------------------------------------------- configs[3]: the 1M + 1M shard --

def test_c4shard_workload_both_sets_and_count_vector(oracle):
    """BASELINE configs[3] per-GPU shard (synth.workload("c4shard"): 125k input + 125k control regions x 500 bp x 579
    PWMs): size-independent properties of each set, a 300-region sample of EACH set bit for bit against the oracle, and the
    int64[2 * 579] vector that feeds the all-reduce recounted from the hits."""
    wl = synth.workload("c4shard")
    vals, widths, cutoffs, P = wl["pwm_values"], wl["widths"], wl["cutoffs"], wl["n_pwms"]
    pw = _lib.PwmSet(vals, widths, cutoffs)
    mr = pw.max_raw()
    count_vector = []
    for s, (bases, offsets) in enumerate(wl["sets"]):
        sq = _lib.SeqSet(bases, offsets)
        res = _lib.scan(pw, sq, 3)
        h = res.hits()
        st = res.stats()
        n = len(h["pos"])
        assert n == st["n_hits"] > 5_000_000 and st["n_pwms_exact"] == 0 and st["n_passes"] == 1
        assert st["n_windows"] == int(sum(500 - int(w) + 1 for w in widths)) * 125_000
        assert order_key_increasing(h, 10)
        assert (h["pos"] >= 0).all() and (h["pos"] + widths[h["motif"]] <= 500).all() and (h["seq_idx"] < 125_000).all()
        assert (h["score"] - cutoffs[h["motif"]] >= -1e-10).all() and (h["score"] * mr[h["motif"]] <= mr[h["motif"]] + 1e-9).all()
        counts = res.region_counts()
        assert np.array_equal(counts, recount_regions(h, P))
        count_vector.append(counts)
        # idempotence
        res2 = _lib.scan(pw, sq, 3)
        h2 = res2.hits(packed=True)                      # ... through the compact copy-out
        assert all(np.array_equal(h[k], h2[k]) for k in ("seq_idx", "pos", "score", "strand", "motif_offsets"))
        res2.close()
        # strand 3 = strand 1 U strand 2
        h1, hr = _lib.scan(pw, sq, 1).hits(), _lib.scan(pw, sq, 2).hits()
        f = h["strand"] == 1
        assert np.array_equal(h["pos"][f], h1["pos"]) and np.array_equal(h["score"][f], h1["score"])
        assert np.array_equal(h["pos"][~f], hr["pos"]) and np.array_equal(h["score"][~f], hr["score"])
        # a 300-region sample from the middle of the set == the oracle, and == the same rows of the full scan
        r0 = 61_000 + 17 * s
        sub_b, sub_o = dist.take_shard(bases, offsets, r0, r0 + 300)
        want = oracle.scan_arrays(vals, widths, cutoffs, sub_b.tobytes(), sub_o, 3, 8)
        m = (h["seq_idx"] >= r0) & (h["seq_idx"] < r0 + 300)
        assert m.sum() == len(want["pos"]) > 0
        assert np.array_equal(h["seq_idx"][m] - r0, want["seq_idx"]) and np.array_equal(h["pos"][m], want["pos"])
        assert np.array_equal(h["score"][m], want["score"]) and np.array_equal(h["strand"][m].astype(np.int32), want["strand"])
        res.close(); sq.close()
    vec = np.concatenate(count_vector)                  # what dist.allreduce_counts sums over the ranks
    assert vec.shape == (2 * P,) and vec.dtype == np.int64 and (vec > 0).all() and (vec <= 125_000).all()
    # the rank-level entry point gives the same vector (batched through a stream, merged)
    out = dist.scan_sharded(vals, widths, cutoffs, wl["sets"], 0, 1, 3)
    assert np.array_equal(out["counts"].ravel(), vec)


def test_c4_full_size_set_as_one_scan(oracle):
    """BASELINE configs[3] at the size bench.py launches at N = 1: ONE 1M x 500 bp set (500 Mbase, ~6e7 hits) as a single
    ms_scan -- order key, ranges, region counts recounted, a 300-region sample from the far end bit for bit against the
    oracle, and the set's first 125k regions equal to the same regions scanned as the 8-GPU shard."""
    wl = synth.c4_shard(0, 1)
    vals, widths, cutoffs, P = wl["pwm_values"], wl["widths"], wl["cutoffs"], wl["n_pwms"]
    bases, offsets = wl["sets"][0]
    R = len(offsets) - 1
    assert R == 1_000_000 and int(offsets[-1]) == 500_000_000
    pw = _lib.PwmSet(vals, widths, cutoffs)
    sq = _lib.SeqSet(bases, offsets)
    res = _lib.scan(pw, sq, 3)
    st = res.stats()
    h = res.hits(copy=False)
    n = len(h["pos"])
    assert n == st["n_hits"] > 50_000_000 and st["n_pwms_exact"] == 0 and st["n_passes"] == 1 and st["n_tiles"] == 1
    assert st["n_windows"] == int(sum(500 - int(w) + 1 for w in widths)) * R
    assert order_key_increasing(h, 10)
    assert (h["pos"] >= 0).all() and (h["pos"] + widths[h["motif"]] <= 500).all() and (h["seq_idx"] >= 0).all() and (h["seq_idx"] < R).all()
    assert (h["score"] - cutoffs[h["motif"]] >= -1e-10).all()
    assert np.array_equal(res.region_counts(), recount_regions(h, P))
    r0 = 987_650
    sub_b, sub_o = dist.take_shard(bases, offsets, r0, r0 + 300)
    want = oracle.scan_arrays(vals, widths, cutoffs, sub_b.tobytes(), sub_o, 3, 8)
    m = (h["seq_idx"] >= r0) & (h["seq_idx"] < r0 + 300)
    assert m.sum() == len(want["pos"]) > 0
    assert np.array_equal(h["seq_idx"][m] - r0, want["seq_idx"]) and np.array_equal(h["pos"][m], want["pos"])
    assert np.array_equal(h["score"][m], want["score"]) and np.array_equal(h["strand"][m].astype(np.int32), want["strand"])
    # the first 125k regions are block 0 of the set = rank 0's shard of 8: the same hits whichever launch size scanned them
    sh_b, sh_o = dist.take_shard(bases, offsets, 0, 125_000)
    hs = _lib.scan(pw, _lib.SeqSet(sh_b, sh_o), 3).hits()
    f = h["seq_idx"] < 125_000
    assert f.sum() == len(hs["pos"])
    assert all(np.array_equal(h[k][f], hs[k]) for k in ("seq_idx", "pos", "score", "strand"))
    res.close(); sq.close()


def test_host_streamed_sweep_with_the_default_span(oracle):
    """BASELINE configs[4] with the product's own span size (375 Mbase): an 800 Mbp genome of 3 chromosomes on the host streamed
    through ms_stream_submit_span -- span plan, properties of every span's result, window counts recounted, and the last 400
    windows of the genome bit for bit against the oracle."""
    window, stride = synth.C5_SHARD["window"], synth.C5_SHARD["stride"]
    lens = [420_000_000, 250_000_000, 130_000_000]
    chroms = []
    for i, L in enumerate(lens):
        b, _ = synth.make_regions(1, L, seed=7100 + i, frac_n=0.0)
        b[L // 2:L // 2 + 20_000] = ord("N")                      # an assembly gap
        b[L - 3000:L - 2950] = ord("n")
        chroms.append(b)
    vals, widths, cutoffs = synth.load_motif_set(579)
    P = len(widths)
    pw = _lib.PwmSet(vals, widths, cutoffs)
    max_span = 375_000_000
    spans = _lib.sweep_spans(lens, window, stride, max_span)
    assert len(spans) == 4 and [s[0] for s in spans] == [0, 0, 1, 2]
    n_win_total = sum((L - window) // stride + 1 for L in lens)
    assert sum(s[4] for s in spans) == n_win_total == spans[-1][3] + spans[-1][4]
    counts = np.zeros(P, dtype=np.int64)
    last = None
    n_sites = 0
    for sp, res in _lib.sweep_stream(pw, chroms, window, stride, max_span, 3, spans=spans, packed=True):
        h = res.hits(copy=False)
        assert sp[2] - sp[1] <= max_span and (sp[4] - 1) * stride + window == sp[2] - sp[1]
        assert len(h["pos"]) == res.n_hits > 10_000_000
        assert order_key_increasing(h, 8)
        assert (h["seq_idx"] >= 0).all() and (h["seq_idx"] < sp[4]).all()
        assert (h["pos"] >= 0).all() and (h["pos"] + widths[h["motif"]] <= window).all()
        rc = res.region_counts()
        assert np.array_equal(rc, recount_regions(h, P))
        counts += rc
        n_sites += res.n_hits
        if sp is spans[-1] or sp == spans[-1]:
            keep = h["seq_idx"] >= sp[4] - 400
            last = {k: h[k][keep].copy() for k in ("seq_idx", "pos", "score", "strand", "motif")}
            last["seq_idx"] -= sp[4] - 400
        res.close()
    assert n_sites > 300_000_000 and (counts > 0).all()
    b = chroms[-1]
    k0 = (len(b) - window) // stride + 1 - 400
    wins = np.concatenate([b[k * stride:k * stride + window] for k in range(k0, k0 + 400)])
    want = oracle.scan_arrays(vals, widths, cutoffs, wins.tobytes(), np.arange(401, dtype=np.int64) * window, 3, 8)
    wm = np.repeat(np.arange(P), np.diff(want["motif_offsets"]))
    assert len(want["pos"]) == len(last["pos"]) > 1000
    assert np.array_equal(last["motif"], wm) and np.array_equal(last["seq_idx"], want["seq_idx"]) and np.array_equal(last["pos"], want["pos"])
    assert np.array_equal(last["score"], want["score"]) and np.array_equal(last["strand"].astype(np.int32), want["strand"])


def test_c5_full_size_three_gbp_host_streamed_sweep(oracle):
    """BASELINE configs[4] AT ITS FULL SIZE: a 3 Gbp / 24-chromosome genome on the host (synth.c5_chrom_lengths / c5_genome's
    blocks), swept as 200 bp windows stride 50 (60M windows) x 579 PWMs through the product's span stream.
      1. the span plan tiles every window of every chromosome exactly once and every base of the genome is scanned once;
      2. counts-only pass over ALL spans (what stats.py:29-31 consumes): per-motif window counts;
      3. a second pass WITH hits over the spans of three chromosomes: order, ranges, and the window counts recounted from the
         hit arrays == the counts-only pass's vectors for those spans;
      4. the last 400 windows of three different chromosomes bit for bit against the oracle
         (window extraction scanner.py:71-87, scoring cscore.c:336-390).
    The genome is generated by a fresh child process (32+ numpy workers writing into /dev/shm files): this process has a GPU
    context and never forks workers itself."""
    import shutil, subprocess, tempfile
    window, stride, max_span = synth.C5["window"], synth.C5["stride"], 375_000_000
    lens = synth.c5_chrom_lengths()
    assert int(lens.sum()) == 3_000_000_000 and len(lens) == 24
    tmp = tempfile.mkdtemp(prefix="ms_c5_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    pins = {}
    try:
        code = ("import sys; sys.path.insert(0, %r); from motifscan_amd import synth; "
                "synth.c5_genome_to_dir(%r, synth.c5_chrom_lengths(), workers=min(48, __import__('os').cpu_count() or 1))" % (ROOT, tmp))
        subprocess.run([sys.executable, "-c", code], check=True, timeout=600)
        chroms = {}
        for ch in range(24):
            pb = _lib.PinnedBuffer(int(lens[ch]))
            pb.array[:] = np.fromfile(os.path.join(tmp, f"chr{ch}.u8"), dtype=np.uint8)
            os.unlink(os.path.join(tmp, f"chr{ch}.u8"))
            pins[ch] = pb
            chroms[ch] = pb.array
        vals, widths, cutoffs = synth.load_motif_set(579)
        P = len(widths)
        pw = _lib.PwmSet(vals, widths, cutoffs)
        # 1. the plan
        spans = _lib.sweep_spans(lens, window, stride, max_span)
        n_win = [(int(L) - window) // stride + 1 for L in lens]
        assert sum(n_win) == 59_999_916 and spans[-1][3] + spans[-1][4] == sum(n_win)           # configs[4]'s 60M windows
        nxt, per_chrom = 0, {}
        for ch, b0, b1, w0, nw in spans:
            assert w0 == nxt and nw > 0 and b1 - b0 <= max_span and (nw - 1) * stride + window == b1 - b0 and b0 % stride == 0
            first_of_chrom = sum(n_win[:ch])
            assert b0 == (w0 - first_of_chrom) * stride                  # the span starts at its first window
            per_chrom[ch] = per_chrom.get(ch, 0) + nw
            nxt += nw
        assert [per_chrom[ch] for ch in range(24)] == n_win
        # every base once: consecutive spans of a chromosome overlap by window - stride, nothing else is uploaded twice
        bases_uploaded = sum(b1 - b0 for _, b0, b1, _, _ in spans)
        assert bases_uploaded == sum((n - 1) * stride + window for n in n_win) + (len(spans) - 24) * (window - stride)
        # 2. counts only, all spans
        counts_by_span, sites_of_span, n_sites_counts, bases_scanned = [], [], 0, 0
        for sp, res in _lib.sweep_stream(pw, chroms, window, stride, max_span, 3, _lib.MS_STREAM_NO_HITS, spans=spans):
            counts_by_span.append(res.region_counts().copy())
            st = res.stats()
            bases_scanned += st["n_bases"]
            n_sites_counts += res.n_hits
            sites_of_span.append(res.n_hits)
            if sp == spans[0]:
                with pytest.raises(ValueError, match="counts-only"):          # a counts-only span hands no site out: asking for them fails loudly
                    res.hits()
            res.close()
        assert len(counts_by_span) == len(spans) and bases_scanned == bases_uploaded
        sites_by_span = dict(zip(spans, sites_of_span))
        total = np.sum(counts_by_span, axis=0)
        assert (total > 0).all() and (total <= sum(n_win)).all()
        # 3. + 4. hits for three chromosomes of different sizes
        picked = [5, 14, 23]
        sub = [sp for sp in spans if sp[0] in picked]
        tails = {}
        for sp, res in _lib.sweep_stream(pw, chroms, window, stride, max_span, 3, spans=sub, packed=True):
            h = res.hits(copy=False)
            assert len(h["pos"]) == res.n_hits > 1_000_000 and order_key_increasing(h, 8)
            assert (h["seq_idx"] >= 0).all() and (h["seq_idx"] < sp[4]).all()
            assert (h["pos"] >= 0).all() and (h["pos"] + widths[h["motif"]] <= window).all()
            assert np.array_equal(recount_regions(h, P), counts_by_span[spans.index(sp)])          # == the counts-only pass
            assert np.array_equal(res.region_counts(), counts_by_span[spans.index(sp)])
            assert res.n_hits == sites_by_span[sp]                                                     # ... and the same number of sites
            if sp[3] + sp[4] == sum(n_win[:sp[0] + 1]):                                                # the chromosome's last span
                keep = h["seq_idx"] >= sp[4] - 400
                t = {k: h[k][keep].copy() for k in ("seq_idx", "pos", "score", "strand", "motif")}
                t["seq_idx"] -= sp[4] - 400
                tails[sp[0]] = t
            res.close()
        assert sorted(tails) == picked
        for ch in picked:
            b = chroms[ch]
            k0 = n_win[ch] - 400
            wins = np.concatenate([b[k * stride:k * stride + window] for k in range(k0, k0 + 400)])
            want = oracle.scan_arrays(vals, widths, cutoffs, wins.tobytes(), np.arange(401, dtype=np.int64) * window, 3, 8)
            wm = np.repeat(np.arange(P), np.diff(want["motif_offsets"]))
            got = tails[ch]
            assert len(want["pos"]) == len(got["pos"]) > 1000
            assert np.array_equal(got["motif"], wm) and np.array_equal(got["seq_idx"], want["seq_idx"]) and np.array_equal(got["pos"], want["pos"])
            assert np.array_equal(got["score"], want["score"]) and np.array_equal(got["strand"].astype(np.int32), want["strand"])
        pw.close()
    finally:
        for pb in pins.values():
            pb.close()
        shutil.rmtree(tmp, ignore_errors=True)


# ------------------------------------------------------------- configs[4]: the sweep shard --

def test_c5shard_sweep_all_579_motifs(oracle):
    """BASELINE configs[4] per-GPU shard (375 Mbp as 200 bp windows, stride 50 = 7.5 M windows, 579 PWMs) through
    ms_scan_sweep: properties of the 1.8e8-site result, the window counts recounted from the sites, and the last 400
    windows bit for bit against the oracle."""
    wl = synth.workload("c5shard")
    vals, widths, cutoffs, P = wl["pwm_values"], wl["widths"], wl["cutoffs"], wl["n_pwms"]
    genome = wl["genome"]
    window, stride = synth.C5_SHARD["window"], synth.C5_SHARD["stride"]
    n_win = wl["n_regions"]
    pw = _lib.PwmSet(vals, widths, cutoffs)
    rg = _lib.ResidentGenome({"chr": genome})
    res = _lib.scan_sweep(pw, rg, "chr", 0, len(genome), window, stride, 3)
    h = res.hits(copy=False)
    n = len(h["pos"])
    assert n == res.stats()["n_hits"] > 100_000_000
    assert res.stats()["n_bases"] == (n_win - 1) * stride + window       # every base scanned once
    assert order_key_increasing(h, 8)
    assert (h["seq_idx"] >= 0).all() and (h["seq_idx"] < n_win).all()
    assert (h["pos"] >= 0).all() and (h["pos"] + widths[h["motif"]] <= window).all()
    assert np.array_equal(res.region_counts(), recount_regions(h, P))
    # every site appears in each of the windows that hold it whole: the same site seen from window k and k+1
    # (genome position = 50 k + pos) -- spot-check on one motif that the multiset of genome positions has the right multiplicity
    mo = h["motif_offsets"]
    p = int(np.argmax(widths == 8))
    a, b = int(mo[p]), int(mo[p + 1])
    gpos = h["seq_idx"][a:b] * stride + h["pos"][a:b]
    interior = (gpos >= window) & (gpos + 8 <= len(genome) - window)
    uniq, mult = np.unique(gpos[interior] * 2 + (h["strand"][a:b][interior] == 2), return_counts=True)
    # a width-8 site at genome position g lies in floor(g / 50) - ceil((g + 8 - 200) / 50) + 1 windows (3 or 4)
    g = uniq >> 1
    expect = g // stride - -((g + 8 - window) // -stride) + 1
    assert np.array_equal(mult, expect)
    n_tail = 400
    r0 = n_win - n_tail
    st = np.arange(r0, n_win, dtype=np.int64) * stride
    tail_bases = np.concatenate([genome[x:x + window] for x in st])
    tail_off = np.arange(n_tail + 1, dtype=np.int64) * window
    want = oracle.scan_arrays(vals, widths, cutoffs, tail_bases.tobytes(), tail_off, 3, 8)
    m = h["seq_idx"] >= r0
    assert m.sum() == len(want["pos"]) > 0
    assert np.array_equal(h["seq_idx"][m] - r0, want["seq_idx"]) and np.array_equal(h["pos"][m], want["pos"])
    assert np.array_equal(h["score"][m], want["score"]) and np.array_equal(h["strand"][m].astype(np.int32), want["strand"])
    del h
    res.close(); rg.close()


# ------------------------------------------------------------------------- batch streams --

@pytest.mark.parametrize("flags,packed", [(0, False), (0, True), (_lib.MS_STREAM_DEDUP, False), (_lib.MS_STREAM_EXACT_ONLY, True)])
def test_stream_batches_equal_single_call_and_oracle(oracle, flags, packed):
    """Batched through ms_stream (upload + pack | scan | copy-out overlapped) == one ms_scan call == the oracle; ragged
    regions with N runs, uneven batch sizes incl. an empty batch, more batches than the stream's capacity."""
    vals, widths, cutoffs = synth.load_motif_set(96)
    bases, offsets = synth.make_regions(5000, 400, seed=21, frac_n=0.05, ragged=True)
    pw = _lib.PwmSet(vals, widths, cutoffs)
    single = _lib.scan(pw, _lib.SeqSet(bases, offsets), 3, _lib.MS_SCAN_EXACT_ONLY if flags & _lib.MS_STREAM_EXACT_ONLY else 0)
    if flags & _lib.MS_STREAM_DEDUP:
        single.dedup(pw)
    want = single.hits()
    cuts = [0, 1, 400, 400, 900, 1700, 1701, 2500, 2600, 3000, 3100, 3600, 3700, 4100, 4200, 4800, 4990, 5000]
    bounds = list(zip(cuts[:-1], cuts[1:]))
    assert len(bounds) > _lib.Stream(pw).capacity
    parts, counts = [], np.zeros(len(widths), dtype=np.int64)
    gen = _lib.scan_stream(pw, (dist.take_shard(bases, offsets, a, b) for a, b in bounds), 3, flags, depth=2, packed=packed)
    for (a, b), res in zip(bounds, gen):
        parts.append((res.hits(packed=packed), a))
        counts += res.region_counts()
        res.close()
    got = _lib.merge_hits(parts, len(widths))
    assert_same_hits(got, want)
    assert np.array_equal(counts, single.region_counts())
    if not flags & _lib.MS_STREAM_DEDUP:
        assert_same_hits(got, oracle.scan_arrays(vals, widths, cutoffs, bases.tobytes(), offsets, 3, 8))


def test_stream_counts_only_errors_and_pinned_input(oracle):
    vals, widths, cutoffs = synth.load_motif_set(40)
    bases, offsets = synth.make_regions(3000, 300, seed=22)
    pw = _lib.PwmSet(vals, widths, cutoffs)
    want = _lib.scan(pw, _lib.SeqSet(bases, offsets), 3)
    pin = _lib.PinnedBuffer(bases.size)                        # page-locked input: the upload overlaps the previous scan
    pin.array[:] = bases
    st = _lib.Stream(pw, 3, _lib.MS_STREAM_NO_HITS, depth=1)
    for a, b in ((0, 1500), (1500, 3000)):
        lo, hi = int(offsets[a]), int(offsets[b])
        st.submit(pin.array[lo:hi], offsets[a:b + 1] - lo)
    c = np.zeros(len(widths), dtype=np.int64)
    n = 0
    while st.in_flight:
        r = st.next()
        c += r.region_counts()
        n += r.n_hits
        r.close()
    assert st.next() is None
    assert np.array_equal(c, want.region_counts()) and n == want.n_hits
    # the reference-shaped job in ONE stream (cli/scan.py:81-89): input batches with their hits out (compact form), control batches
    # counts only (ms_stream_submit_counts_only) -- same counts, same hits for the batches that carry them
    half = int(offsets[1500])
    mixed = [(pin.array[:half], offsets[:1501], False), (pin.array[half:], offsets[1500:] - half, True),
             (pin.array[:half], offsets[:1501], True), (pin.array[half:], offsets[1500:] - half, False)]
    got = list(_lib.scan_stream(pw, iter(mixed), 3, packed=True))
    wa = _lib.scan(pw, _lib.SeqSet(bases[:half], offsets[:1501]), 3)
    wb = _lib.scan(pw, _lib.SeqSet(bases[half:], offsets[1500:] - half), 3)
    for r, w, counts_only in zip(got, (wa, wb, wa, wb), (False, True, True, False)):
        assert r.n_hits == w.n_hits and np.array_equal(r.region_counts(), w.region_counts())
        assert np.array_equal(r.motif_offsets, w.motif_offsets)                    # a counts-only batch still knows every motif's number of sites
        if counts_only:
            with pytest.raises(ValueError):                                        # ... but holds no site arrays (counted unordered: MS_SCAN_COUNTS_ONLY)
                r.hits()
        if not counts_only:
            hp, hw = r.hits(packed=True), w.hits()
            for k in ("motif_offsets", "seq_idx", "pos", "score"):
                assert np.array_equal(hp[k], hw[k])
        r.close()
    wa.close(); wb.close()
    # a bad batch fails at next(), in order, with the library's message; the stream stays usable
    st.submit(bases[:600], np.array([0, 300, 600], dtype=np.int64))
    with pytest.raises(ValueError):
        st.submit(bases[:10], np.array([0, 20], dtype=np.int64))           # offsets past the buffer: refused at submit
    bad = np.array([0, 400, 300, 600], dtype=np.int64)                      # not monotone: refused by the uploader
    _lib.check(_lib.lib().ms_stream_submit(st.h, bases.ctypes.data, _lib.ptr(bad, _lib.ctypes.c_int64), 3))
    st._keep.append(bases)
    ok = st.next()
    assert ok.n_hits >= 0
    ok.close()
    with pytest.raises(ValueError, match="non-decreasing"):
        st.next()
    st.close(); pin.close()
    with pytest.raises(ValueError):
        _lib.Stream(pw, 3, 0, depth=0)


@pytest.mark.parametrize("shape", ["ragged", "one_long_region", "single_strand", "empty_and_tiny"])
def test_counts_only_scan_equals_the_ordered_scan(oracle, shape):
    """MS_SCAN_COUNTS_ONLY (round 6): n_hits, per-motif site numbers and per-motif region counts from the UNORDERED hits (a bit per
    (motif, region), no radix sort, no finalize) == the same three from the ordered scan and from the oracle's nested result
    (stats.py:29-31: sum(len(sites_by_region) > 0 ...)).  Where the bitmap form does not apply (a region so long that the hit keys carry
    global positions) the scan falls back to the ordered path by itself: same numbers."""
    vals, widths, cutoffs = synth.load_motif_set(90)
    pw = _lib.PwmSet(vals, widths, cutoffs)
    strand = 3
    if shape == "ragged":
        bases, offsets = synth.make_regions(4000, 260, seed=41, frac_n=0.06, ragged=True)
    elif shape == "one_long_region":
        b1, o1 = synth.make_regions(300, 120, seed=42)
        b2, _ = synth.make_regions(1, 900_000, seed=43)
        bases = np.concatenate([b1, b2, b1])
        offsets = np.concatenate([o1, o1[-1] + np.array([b2.size]), o1[-1] + b2.size + o1[1:]]).astype(np.int64)
    elif shape == "single_strand":
        bases, offsets = synth.make_regions(2500, 500, seed=44, frac_n=0.02)
        strand = 2
    else:
        lens = np.array([0, 3, 0, 500, 1, 499, 0, 0, 64, 63, 65, 0])
        rng = np.random.default_rng(45)
        bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(lens.sum()))]
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    sq = _lib.SeqSet(bases, offsets)
    full = _lib.scan(pw, sq, strand)
    want = oracle.scan_arrays(vals, widths, cutoffs, bases.tobytes(), offsets, strand, 8)
    pair = np.unique((np.repeat(np.arange(len(widths)), np.diff(want["motif_offsets"])).astype(np.int64) << 32) | want["seq_idx"])
    want_regions = np.bincount(pair >> 32, minlength=len(widths))
    for rep in range(2):                                     # (the second scan takes the predicted-size, one-sync form)
        res = _lib.scan(pw, sq, strand, _lib.MS_SCAN_COUNTS_ONLY)
        assert res.n_hits == full.n_hits == len(want["pos"])
        assert np.array_equal(res.motif_offsets, want["motif_offsets"])
        assert np.array_equal(res.region_counts(), want_regions) and np.array_equal(full.region_counts(), want_regions)
        # the flag map is taken where the set holds a hit per ~50 (motif, region) cells or more, and region-local keys exist; else the ordered path
        # gives the same three numbers (and its result still holds the site arrays)
        if shape in ("ragged", "single_strand"):
            with pytest.raises(ValueError):
                res.hits()
        elif shape == "one_long_region":
            assert res.hits()["pos"].size == res.n_hits
        res.close()
    full.close(); sq.close(); pw.close()


def test_twelve_byte_copy_out_where_it_fits_sixteen_where_not(oracle):
    """MS_STREAM_PACKED12 / ms_result_hits_packed12_host (VERDICT r5 #2a): coord32 = region << shift | pos << 1 | strand + the fp64 score --
    12 bytes per hit on the host link -- for every batch whose region indices and positions fit 31 bits; a batch that does not fit comes
    out in the 16-byte form by itself (ms_result_packed_form says which).  Both forms decode to the oracle's hits; a result that holds
    one form refuses the other's accessor; a single scan (no stream) can be read in either."""
    vals, widths, cutoffs = synth.load_motif_set(60)
    pw = _lib.PwmSet(vals, widths, cutoffs)
    b1, o1 = synth.make_regions(900, 400, seed=5, frac_n=0.04, ragged=True)             # 10 + 9 + 1 bits: fits
    b2, o2 = synth.make_regions(3, 70000, seed=6, frac_n=0.01)                           # fits (2 + 17 + 1)
    lens = np.r_[np.full(70000, 1), [40000]]                                             # 17 bits of region index + 16 of position: does not fit
    rng = np.random.default_rng(8)
    b3 = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(lens.sum()))]
    o3 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    batches = [(b1, o1), (b2, o2), (b3, o3)]
    forms = []
    for (b_, o_), res in zip(batches, _lib.scan_stream(pw, iter(batches), 3, packed=12)):
        forms.append(res.packed_form())
        got = res.hits(packed=True)
        want = oracle.scan_arrays(vals, widths, cutoffs, b_.tobytes(), o_, 3, 8)
        for k in ("motif_offsets", "seq_idx", "pos", "score"):
            assert np.array_equal(got[k], want[k]), (forms, k)
        assert np.array_equal(got["strand"].astype(np.int32), want["strand"])
        with pytest.raises(ValueError):
            res.hits(packed=16 if forms[-1] == 12 else 12)                               # the form the stream did not make for this batch
        res.close()
    assert forms == [12, 12, 16]
    # the 16-byte stream is unchanged
    for (b_, o_), res in zip(batches[:1], _lib.scan_stream(pw, iter(batches[:1]), 3, packed=True)):
        assert res.packed_form() == 16 and res.hits(packed=True)["pos"].size == res.n_hits
        res.close()
    # a plain scan: either form on demand, and back to the plain arrays
    sq = _lib.SeqSet(b1, o1)
    res = _lib.scan(pw, sq, 3)
    want = oracle.scan_arrays(vals, widths, cutoffs, b1.tobytes(), o1, 3, 8)
    for form in (12, 16, 12):
        got = res.hits(packed=form)
        assert res.packed_form() == form
        assert all(np.array_equal(got[k], want[k]) for k in ("seq_idx", "pos", "score"))
    assert np.array_equal(res.hits()["pos"], want["pos"])
    res.dedup(pw)
    got = res.hits(packed=12)
    assert got["pos"].size == res.n_hits and np.array_equal(got["pos"], res.hits()["pos"])
    res.close(); sq.close(); pw.close()


def test_stream_packs_on_the_scan_stage_and_the_old_form_agrees(oracle, monkeypatch):
    """Round 6: a batch stream's upload stage only copies (ASCII + offsets on the DMA engines); pack_kernel / blk2reg_kernel run on the scan
    stream in front of the batch's pre-filter (seqset_create_upload_only / seqset_pack_pending).  Same results as round 5's form (packing on
    the upload stream, MS_MEASURE=1 MS_STREAM_PACK_IN_UPLOAD=1) and as the oracle, for plain batches, counts-only batches and sweep spans."""
    vals, widths, cutoffs = synth.load_motif_set(80)
    pw = _lib.PwmSet(vals, widths, cutoffs)
    bases, offsets = synth.make_regions(2400, 300, seed=12, frac_n=0.05, ragged=True)
    cuts = [0, 300, 301, 1500, 2400]
    batches = [(bases[int(offsets[a]):int(offsets[b])], offsets[a:b + 1] - offsets[a], k == 2) for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))]
    want = oracle.scan_arrays(vals, widths, cutoffs, bases.tobytes(), offsets, 3, 8)
    outs = {}
    for mode in ("scan_stage", "upload_stage"):
        if mode == "upload_stage":
            monkeypatch.setenv("MS_MEASURE", "1")
            monkeypatch.setenv("MS_STREAM_PACK_IN_UPLOAD", "1")
        parts, counts = [], np.zeros(len(widths), dtype=np.int64)
        for (b_, o_, co), (a, _), res in zip(batches, zip(cuts[:-1], cuts[1:]), _lib.scan_stream(pw, iter(batches), 3, packed=12)):
            counts += res.region_counts()
            if not co:
                parts.append((res.hits(packed=True), a))
            res.close()
        outs[mode] = (_lib.merge_hits(parts, len(widths)), counts)
    monkeypatch.undo()
    sel = (want["seq_idx"] < 301) | (want["seq_idx"] >= 1500)
    for mode, (merged, counts) in outs.items():
        for k in ("seq_idx", "pos", "score"):
            assert np.array_equal(merged[k], want[k][sel]), (mode, k)
        pair = np.unique((np.repeat(np.arange(len(widths)), np.diff(want["motif_offsets"])).astype(np.int64) << 32) | want["seq_idx"])
        assert np.array_equal(counts, np.bincount(pair >> 32, minlength=len(widths)))
    # a sweep span through the stream (kind 1) takes the same upload-only path
    g, _ = synth.make_regions(1, 50000, seed=3, frac_n=0.02)
    spans = list(_lib.sweep_stream(pw, [g], 200, 50, 20000, 3))
    n_sites = sum(r.n_hits for _, r in spans)
    rg = _lib.ResidentGenome({"c": g})
    ref = _lib.scan_sweep(pw, rg, 0, 0, 50000, 200, 50, 3)
    assert n_sites == ref.n_hits and n_sites > 100
    for _, r in spans:
        r.close()
    ref.close(); rg.close(); pw.close()


def test_host_packed_sets_and_streams_equal_device_packed_ones(oracle):
    """ms_seqset_create_hostpacked / MS_STREAM_HOST_PACK: convert_seq and the region hints made by host threads, no kernel in the upload
    stage.  The set scans to the same hits as the device-packed one (ragged regions incl. empty ones, runs of N, lower case, IUPAC letters,
    an unaligned tail), and a stream of host-packed batches -- hits out and counts-only mixed -- equals the oracle."""
    vals, widths, cutoffs = synth.load_motif_set(120)
    pw = _lib.PwmSet(vals, widths, cutoffs)
    rng = np.random.default_rng(77)
    bases, offsets = synth.make_regions(3001, 333, seed=31, frac_n=0.08, ragged=True)
    bases = bases.copy()
    for ch in b"RYKMSWryn-*":                                         # letters that "add nothing" (cscore.c:92-111)
        bases[rng.integers(0, bases.size, 40)] = ch
    lens = np.diff(offsets)
    lens[[5, 6, 900]] = 0                                             # empty regions
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    bases = bases[:int(offsets[-1])]
    want = oracle.scan_arrays(vals, widths, cutoffs, bases.tobytes(), offsets, 3, 8)
    for threads in (1, 5):
        sq = _lib.SeqSet(bases, offsets, host_pack_threads=threads)
        res = _lib.scan(pw, sq, 3)
        assert_same_hits(res.hits(), want)
        res.close(); sq.close()
    cuts = [0, 1, 700, 700, 2200, 3001]
    batches = [(bases[int(offsets[a]):int(offsets[b])], offsets[a:b + 1] - offsets[a], k == 3) for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))]
    parts, st = [], {}
    results = list(_lib.scan_stream(pw, iter(batches), 3, packed=True, host_pack=True, stage_stats=st))
    for (b_, o_, counts_only), (a, _), res in zip(batches, zip(cuts[:-1], cuts[1:]), results):
        ref = _lib.scan(pw, _lib.SeqSet(b_, o_), 3)
        assert res.n_hits == ref.n_hits and np.array_equal(res.region_counts(), ref.region_counts())
        if not counts_only:
            parts.append((res.hits(packed=True), a))
        res.close(); ref.close()
    merged = _lib.merge_hits(parts, len(widths))
    # (the counts-only batch is regions [700, 2200): its hits stay on the device)
    sel = (want["seq_idx"] < 700) | (want["seq_idx"] >= 2200)
    wm = np.repeat(np.arange(len(widths)), np.diff(want["motif_offsets"]))
    for k in ("seq_idx", "pos", "score"):
        assert np.array_equal(merged[k], want[k][sel])
    assert np.array_equal(np.repeat(np.arange(len(widths)), np.diff(merged["motif_offsets"])), wm[sel])
    assert st["upload"]["batches"] == len(batches)
    pw.close()


def test_stream_stage_clocks_and_block_pool_statistics():
    """ms_stream_stats counts every batch once per stage and its clocks are sane; ms_device_pool_stats: a second pass over the
    same batches is served from the block cache (size classes), not by the driver."""
    vals, widths, cutoffs = synth.load_motif_set(40)
    bases, offsets = synth.make_regions(4000, 300, seed=23, ragged=True)
    pw = _lib.PwmSet(vals, widths, cutoffs)
    bounds = [(0, 700), (700, 1900), (1900, 2000), (2000, 4000)]

    def one_pass():
        st = {}
        n = 0
        for res in _lib.scan_stream(pw, (dist.take_shard(bases, offsets, a, b) for a, b in bounds), 3, 0, depth=2, packed=True,
                                    stage_stats=st):
            n += res.n_hits
            res.close()
        return n, st

    n1, _ = one_pass()
    p0 = _lib.pool_stats()
    n2, st = one_pass()
    p1 = _lib.pool_stats()
    assert n1 == n2 == _lib.scan(pw, _lib.SeqSet(bases, offsets), 3).n_hits
    assert set(st) == {"upload", "scan", "copy_out"}
    for stage in st.values():
        assert stage["batches"] == len(bounds)
        assert stage["ms_work"] >= 0 and stage["ms_wait_in"] >= 0 and stage["ms_wait_out"] >= 0
    assert st["upload"]["ms_work"] > 0 and st["scan"]["ms_work"] > 0 and st["copy_out"]["ms_work"] > 0
    assert p1["hits"] > p0["hits"]
    assert p1["misses"] == p0["misses"], "the second pass went to hipMalloc: %r -> %r" % (p0, p1)
    assert p1["cached_bytes"] > 0 and p1["cached_blocks"] > 0


def test_cu_partitioned_streams_give_the_same_result():
    """MS_MEASURE=1 MS_CU_PARTITION=1 (CU-masked copy / scan streams while a batch stream is live; read once per process, hence
    the child process): batched results == the single call, and the resident path of the same process is untouched."""
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from motifscan_amd import _lib, synth, dist
vals, widths, cutoffs = synth.load_motif_set(64)
bases, offsets = synth.make_regions(6000, 400, seed=29, frac_n=0.03, ragged=True)
pw = _lib.PwmSet(vals, widths, cutoffs)
want = _lib.scan(pw, _lib.SeqSet(bases, offsets), 3).hits()
bounds = [(0, 1000), (1000, 1001), (1001, 3500), (3500, 6000)]
parts = []
for (a, b), res in zip(bounds, _lib.scan_stream(pw, (dist.take_shard(bases, offsets, a, b) for a, b in bounds), 3, 0, depth=2)):
    parts.append((res.hits(), a)); res.close()
got = _lib.merge_hits(parts, len(widths))
again = _lib.scan(pw, _lib.SeqSet(bases, offsets), 3).hits()          # after the stream is gone: whole-device streams again
for k in ("motif_offsets", "seq_idx", "pos", "strand"):
    assert np.array_equal(got[k], want[k]) and np.array_equal(again[k], want[k]), k
assert np.array_equal(got["score"].view(np.uint64), want["score"].view(np.uint64))
print("OK", int(want["motif_offsets"][-1]))
""" % ROOT
    env = dict(os.environ, MS_MEASURE="1", MS_CU_PARTITION="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert out.stdout.strip().splitlines()[-1].startswith("OK ")


@pytest.mark.parametrize("n_regions", [43_500, 44_800, 46_500])
def test_work_handout_regimes_agree_with_the_all_fp64_kernel(n_regions):
    """The pre-filter hands its work out per wave: an input worth fewer than 8 units per wave is split evenly with no atomics, a
    larger one goes through the tile's counter word (8 passes of 64 window starts per unit on the 579-motif set, i.e. the
    switch sits near 16.8 Mbase = 44 700 ragged regions of 250-500 bp).  Sizes either side of it, ragged region lengths and a last region that ends mid-pass:
    the result must equal the all-fp64 kernel's (MS_SCAN_EXACT_ONLY: no pre-filter at all), hit for hit and bit for bit."""
    vals, widths, cutoffs = synth.load_motif_set(579)
    bases, offsets = synth.make_regions(n_regions, 500, seed=31, frac_n=0.01, ragged=True)
    cut = int(offsets[-1]) - 37                                # drop 37 bases: the last region ends inside a pass
    offsets = offsets.copy(); offsets[-1] = cut
    bases = bases[:cut]
    pw = _lib.PwmSet(vals, widths, cutoffs)
    sq = _lib.SeqSet(bases, offsets)
    got = _lib.scan(pw, sq, 3)
    st = got.stats()
    assert st["pf_engine"] == 3 and st["n_tiles"] == 1
    want = _lib.scan(pw, sq, 3, _lib.MS_SCAN_EXACT_ONLY)
    assert_same_hits(got.hits(), want.hits())
    assert np.array_equal(got.region_counts(), want.region_counts())


@pytest.mark.parametrize("n_motifs,strand", [(50, 3), (579, 1), (1700, 3), (2400, 2)])
def test_work_handout_over_sizes_and_tiles_against_the_all_fp64_kernel(n_motifs, strand):
    """The hand-out's unit size and regime depend on the motif set (k-blocks per tile, number of LDS tiles) and on the input size;
    a sweep over both, single strands included, against the kernel that uses no pre-filter at all."""
    vals, widths, cutoffs = synth.load_motif_set(min(n_motifs, 579))
    if n_motifs > 579:                                         # the set repeated: several LDS tiles
        mats = synth.matrices_of(vals, widths)
        pick = [i % 579 for i in range(n_motifs)]
        vals = np.concatenate([mats[i].ravel() for i in pick]); widths = widths[pick]; cutoffs = cutoffs[pick]
    pw = _lib.PwmSet(vals, widths, cutoffs)
    for n_regions, length in ((1, 37), (3, 64), (700, 300), (9_000, 500), (40_000, 450)):
        bases, offsets = synth.make_regions(n_regions, length, seed=100 + n_regions, frac_n=0.02, ragged=n_regions > 3)
        sq = _lib.SeqSet(bases, offsets)
        got = _lib.scan(pw, sq, strand)
        want = _lib.scan(pw, sq, strand, _lib.MS_SCAN_EXACT_ONLY)
        assert_same_hits(got.hits(), want.hits())
        assert np.array_equal(got.region_counts(), want.region_counts())
        if n_motifs == 1700:                                   # (1300 motifs still fit one LDS tile)
            assert got.stats()["n_tiles"] >= 2
        got.close(); want.close(); sq.close()


@pytest.mark.parametrize("max_blocks", [1, 3, 5, 8, 13])
def test_work_handout_with_fewer_blocks_than_counter_words(monkeypatch, max_blocks):
    """A very large motif set leaves a tile only a few blocks; the hand-out deals the blocks onto min(8, blocks) counter words --
    a word without a block would never hand its units out.  MS_PF_MAX_BLOCKS (a measurement switch) caps the blocks per tile."""
    monkeypatch.setenv("MS_MEASURE", "1")
    monkeypatch.setenv("MS_PF_MAX_BLOCKS", str(max_blocks))
    vals, widths, cutoffs = synth.load_motif_set(120)
    bases, offsets = synth.make_regions(6000, 500, seed=77, frac_n=0.01, ragged=True)
    pw = _lib.PwmSet(vals, widths, cutoffs)
    sq = _lib.SeqSet(bases, offsets)
    got = _lib.scan(pw, sq, 3)
    want = _lib.scan(pw, sq, 3, _lib.MS_SCAN_EXACT_ONLY)
    assert got.n_hits > 1000
    assert_same_hits(got.hits(), want.hits())


def test_owned_views_keep_the_result_alive():
    """views of the library's pinned buffers must not dangle when the caller drops the result object."""
    import gc
    vals, widths, cutoffs = synth.load_motif_set(30)
    bases, offsets = synth.make_regions(2000, 300, seed=23)
    pw = _lib.PwmSet(vals, widths, cutoffs)
    sq = _lib.SeqSet(bases, offsets)
    h = _lib.scan(pw, sq, 3).hits(copy=False)              # the ScanResult itself is dropped right here
    keep = h["score"][10:200]                              # a slice of a view
    keep2 = np.asarray(h["pos"])                           # numpy collapses .base past any ndarray subclass in between
    snap, snap2 = keep.copy(), keep2.copy()
    del h
    gc.collect()
    for _ in range(4):                                      # scans that would reuse a freed pinned block
        _lib.scan(pw, sq, 1).hits(copy=False)
        gc.collect()
    assert np.array_equal(keep, snap) and np.array_equal(keep2, snap2)
    pin = _lib.PinnedBuffer(1 << 16)
    view = np.asarray(pin.array)[100:200]
    view[:] = 7
    del pin
    gc.collect()
    other = _lib.PinnedBuffer(1 << 16)                      # would take the freed block
    other.array[:] = 1
    assert (view == 7).all()


# --------------------------------------------------------- host-streamed multi-chromosome sweep --

@pytest.mark.parametrize("window,stride,max_span", [(200, 50, 3000), (64, 64, 1000), (30, 7, 500)])
def test_host_streamed_multi_chromosome_sweep_equals_oracle(oracle, jaspar579, window, stride, max_span):
    """BASELINE configs[4] proper: a genome of several chromosomes on the HOST, swept in spans of bounded size through an
    ms_stream (spans overlap by window - stride, window indices global over spans and chromosomes, scanner.py:71-87
    per window) == the oracle over the same windows as separate regions == the resident-genome sweep per chromosome."""
    rng = np.random.default_rng(31)
    sel = rng.choice(579, size=120, replace=False)
    mats = synth.matrices_of(jaspar579["pwm_values"], jaspar579["widths"])
    vals = np.concatenate([mats[i].ravel() for i in sel])
    widths = jaspar579["widths"][sel]
    cutoffs = jaspar579["cutoffs"]["1e-3"][sel]
    lens = [7000, 150, 9000, window - 1, 4100, window]
    chroms = []
    for i, L in enumerate(lens):
        b, _ = synth.make_regions(1, L, seed=40 + i, frac_n=0.0)
        if L > 600:
            b[300:340] = ord("N")
            b[L - 90:L - 60] = ord("n")
        chroms.append(b)
    spans = _lib.sweep_spans(lens, window, stride, max_span)
    assert len(spans) >= 8 and len({s[0] for s in spans}) >= 3
    pw = _lib.PwmSet(vals, widths, cutoffs)
    parts, counts = [], np.zeros(len(widths), dtype=np.int64)
    for sp, res in _lib.sweep_stream(pw, chroms, window, stride, max_span, 3, spans=spans):
        assert sp[2] - sp[1] <= max_span and (sp[4] - 1) * stride + window == sp[2] - sp[1]
        parts.append((res.hits(), sp[3]))
        counts += res.region_counts()
        res.close()
    got = _lib.merge_hits(parts, len(widths))
    # the same sweep counts-only (what stats.py:29-31 reads of it): per-motif window counts and the number of sites of every span equal
    # those of the span's full hand-out -- made by one pass over the span's hit positions, no site is written
    per_span = [(int(len(h["pos"])), off) for h, off in parts]
    counts_only = np.zeros(len(widths), dtype=np.int64)
    for (sp, res), (n_sites, _) in zip(_lib.sweep_stream(pw, chroms, window, stride, max_span, 3, _lib.MS_STREAM_NO_HITS, spans=spans), per_span):
        assert res.n_hits == n_sites
        counts_only += res.region_counts()
        res.close()
    assert np.array_equal(counts_only, counts)
    # the oracle over every window as a region of its own
    win_bases, n_win = [], 0
    for b in chroms:
        for k in range((len(b) - window) // stride + 1 if len(b) >= window else 0):
            win_bases.append(b[k * stride:k * stride + window])
            n_win += 1
    assert n_win == sum(s[4] for s in spans) == spans[-1][3] + spans[-1][4]
    want = oracle.scan_arrays(vals, widths, cutoffs, np.concatenate(win_bases).tobytes(),
                              np.arange(n_win + 1, dtype=np.int64) * window, 3, 8)
    assert len(want["pos"]) > 1000
    assert_same_hits(got, want)
    assert np.array_equal(counts, recount_regions(want, len(widths)))
    # the same through the resident genome, chromosome by chromosome
    rg = _lib.ResidentGenome({f"c{i}": b for i, b in enumerate(chroms)})
    parts2, first = [], 0
    for i, b in enumerate(chroms):
        r = _lib.scan_sweep(pw, rg, f"c{i}", 0, len(b), window, stride, 3)
        parts2.append((r.hits(), first))
        first += (len(b) - window) // stride + 1 if len(b) >= window else 0
    assert_same_hits(_lib.merge_hits(parts2, len(widths)), want)
    # a rank's share of the spans: contiguous, complete, disjoint
    shares = [dist.span_shard(spans, r, 3) for r in range(3)]
    assert sum(shares, []) == spans and all(shares)
    # a sweep that stopped after k spans is resumed from span k (the span list is a pure function of the chromosome lengths, a span's
    # result carries its first window): the two runs' parts together are the whole sweep, in a fresh stream and a fresh PWM handle
    k = len(spans) // 2
    parts3 = list(parts[:k])
    pw2 = _lib.PwmSet(vals, widths, cutoffs)
    for sp, res in _lib.sweep_stream(pw2, chroms, window, stride, max_span, 3, spans=spans[k:]):
        parts3.append((res.hits(), sp[3]))
        res.close()
    assert_same_hits(_lib.merge_hits(parts3, len(widths)), want)


def test_sweep_hands_out_more_than_2_to_the_32_sites_arithmetic():
    """The hand-out's prefix sums are 64-bit: exercised at a size the box can hold (a dense-hit sweep whose site count
    needs > 32 bits cannot be checked here), so check the widened path by a sweep with window / stride = 64."""
    vals, widths, _ = synth.load_motif_set(12)
    cut2 = np.load(synth.MOTIF_SET)["cutoffs"][:12, 0]      # p = 1e-2: dense hits
    genome, _ = synth.make_regions(1, 3_000_000, seed=33, frac_n=0.0)
    pw = _lib.PwmSet(vals, widths, cut2)
    rg = _lib.ResidentGenome({"chr": genome})
    res = _lib.scan_sweep(pw, rg, "chr", 0, len(genome), 256, 4, 3)
    span = _lib.scan(pw, rg.extract([0], [0], [len(genome)]), 3)
    hs = span.hits()
    lo = np.maximum(0, -((hs["pos"] + widths[hs["motif"]] - 256) // -4))
    hi = np.minimum(hs["pos"] // 4, (len(genome) - 256) // 4)
    assert res.n_hits == int(np.maximum(hi - lo + 1, 0).sum()) > 20_000_000


# ----------------------------------------------------- rows either side, fed from the device --

def test_n4_writers_from_device_tables_byte_identical(small, tmp_path):
    """N4 on the GPU box: Scanner.scan_motifs_arrays(with_tables=True) -> write_sites_table / _bed / _enrich_table ==
    what the reference's io module wrote (io/__init__.py:12-71), byte for byte."""
    d = small["N4"]
    for rel, text in d["files"].items():
        path = tmp_path / rel
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_text(text)
    w = d["writers"]
    chroms = small["G2"]["chroms"]
    pwms = formats.read_motifscan_pwms(tmp_path / "test/test_pwms.motifscan")

    class Reg:
        def __init__(self, c, s, e):
            self.chrom, self.start, self.end, self.summit = c, s, e, (s + e) // 2

    def scan(rows, resident):
        regs = [Reg(*r) for r in rows]
        genome = _lib.ResidentGenome(chroms, keep_host=True) if resident else type("G", (), {
            "chrom_sizes": {k: len(v) for k, v in chroms.items()},
            "fetch_sequence": staticmethod(lambda c, s, e: chroms[c][s:e])})
        sc = scanner.Scanner(genome, regs, window_size=0, strand="both", p_value=w["p_value"], remove_dup=True)
        return regs, sc.scan_motifs_arrays(pwms, with_tables=True)

    for resident in (False, True):
        regs, a = scan(w["regions"], resident)
        out = tmp_path / f"out{int(resident)}"
        formats.write_sites_table(out, pwms, regs, a["n_sites"], a["max_score"])
        hits = {"motif": a["motif"], "region": a["region"], "start": a["start"], "score": a["score"], "strand": a["strand"],
                "motif_offsets": a["motif_offsets"]}
        formats.write_sites_bed(out, pwms, regs, hits)
        assert (out / "motif_sites_number.xls").read_text() == w["motif_sites_number.xls"]
        assert (out / "motif_sites_score.xls").read_text() == w["motif_sites_score.xls"]
        assert {f: (out / "motif_sites" / f).read_text() for f in os.listdir(out / "motif_sites")} == w["bed"]
        cregs, ca = scan(w["control_regions"], resident)
        rows = dist.enrichment(a["n_regions_with_site"], ca["n_regions_with_site"], len(regs), len(cregs))
        formats.write_enrich_table(out, [p.matrix_id + "," + p.name for p in pwms], rows)
        assert (out / "motif_enrichment.xls").read_text() == w["motif_enrichment.xls"]
        assert np.array_equal(a["n_regions_with_site"], (a["n_sites"] > 0).sum(axis=1))


def test_make_motif_sites_on_device_c_scan_motif_output(small):
    """a9: scanner.make_motif_sites (scanner.py:135-153) applied to cscore.c_scan_motif's pooled hits == the reference
    Scanner's nested lists on the toy genome (G2), with and without de-duplication (scanner.py:171-193)."""
    g = small["G2"]
    seq = g["chroms"]["chr1"][1:5]                                     # chr1:2-5, window 4 -> 'aTtC' (tests/test_scanner.py:29-54)
    m = [[[1, 0], [0, 1], [0, 0], [1, 0]]]
    for cutoff, n_raw in ((1.0, 1), (0.5, 5)):
        sites = cscore.c_scan_motif(m, [cutoff], [seq], 3, 1)
        assert len(sites[0]) == n_raw
        ms = scanner.make_motif_sites(sites, [1])
        assert [len(x) for x in ms[0]] == [n_raw]
        assert all(isinstance(s, scanner.MotifSite) and s.strand in "+-" for s in ms[0][0])
        assert [s.start for s in ms[0][0]] == [1 + hit[1] for hit in sites[0]]
        dd = scanner.deduplicate_motif_sites(ms, [2])
        if cutoff == 1.0:
            assert [tuple(s) for s in dd[0][0]] == [(3, 1.0, "+")]
        else:
            assert [tuple(s) for s in dd[0][0]] == [(1, .5, "+"), (1, .5, "-"), (3, 1.0, "+")]
    # the same through the Scanner's own nested-list path on a multi-region list
    rows = [["chr1", 0, 10], ["chr2", 0, 17], ["chrX", 0, 16]]
    seqs = [g["chroms"][c][s:e] for c, s, e in rows]
    pw = small["N4"]["pwms"]
    mats = [p["matrix"] for p in pw]
    cuts = [p["cutoffs"]["1e-2"] for p in pw]
    pooled = cscore.c_scan_motif(mats, cuts, seqs, 3, 1)
    nested = scanner.make_motif_sites(pooled, [r[1] for r in rows])
    flat = [(mi, ri, s.start, s.score, s.strand) for mi, per in enumerate(nested) for ri, ss in enumerate(per) for s in ss]
    assert flat == [(mi, h[0], rows[h[0]][1] + h[1], h[2], "+" if h[3] == 1 else "-") for mi, per in enumerate(pooled) for h in per]
    assert len(flat) >= 2


# ------------------------------------------------------------------- N > 1 code path --

def test_bench_self_launches_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` starts its own two ranks (fresh processes, before anything touches the GPU) and reports
    n_gpus = 2.  RCCL refuses two ranks on one device, so on this 1-GPU box the ranks talk over gloo (MS_BENCH_BACKEND, a test
    aid); everything else -- sharding of the full workload, per-rank scans through the library, the all-reduce of the
    device-resident count vector, max-over-ranks timing -- is the N > 1 path the driver runs over RCCL."""
    env = dict(os.environ, MS_BENCH_BACKEND="gloo", MS_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--workload", "tiny", "--no-cpu-baseline", "--min-warm-seconds", "0"], env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["counts_check"]["allreduce_equals_sum_of_rank_counts"] is True
    # the N > 1 line verifies itself: every rank's own step time, the collective's own time, and a sample of every rank's
    # own shard compared with the oracle
    rk = line["ranks"]
    assert len(rk["ms_per_step_by_rank"]) == 2 and 0 < rk["ms_per_step_min"] <= rk["ms_per_step_max"] <= line["ms_per_step"] * 1.5
    assert len(rk["allreduce_ms_mean_by_rank"]) == 2 and all(x >= 0 for x in rk["allreduce_ms_mean_by_rank"])
    ps = rk["parity_sample"]
    assert ps["ranks"] == 2 and ps["ranks_identical_to_oracle"] == 2 and ps["regions_per_rank"] > 0 and ps["hits_checked"] > 0
    assert line["roofline"]["kernel"] == "prefilter_f6_kernel"


def _bench_ranks_on_one_gpu(args, timeout=1500):
    env = dict(os.environ, MS_BENCH_BACKEND="gloo", MS_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_world_of_eight_on_one_gpu():
    """The shape the driver's 8-GPU SCALE run has -- eight fresh ranks, eight shards of both region sets of configs[3] (here 160k
    regions per set), 8 stream pipelines (24 host threads) and 8 block pools, one all-reduce per step -- on the one GPU of this box
    over gloo: n_gpus = 8, the all-reduced vector == the sum of the ranks' vectors, the eight shards tile both sets in rank order,
    and 2000 regions of EVERY rank's own shard are bit-identical to the oracle."""
    line = _bench_ranks_on_one_gpu(["--gpus", "8", "--regions-per-set", "160000", "--steps", "3", "--warmup", "1", "--min-warm-seconds", "0"])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["value"] > 0
    assert line["counts_check"]["allreduce_equals_sum_of_rank_counts"] is True
    rk = line["ranks"]
    assert rk["shards_tile_every_set"] is True and len(rk["shard_by_rank"]) == 8
    assert [b - a for a, b in rk["shard_by_rank"]] == [20_000] * 8
    ps = rk["parity_sample"]
    assert ps["ranks"] == 8 and ps["ranks_identical_to_oracle"] == 8 and ps["regions_per_rank"] == 2000 and ps["hits_checked"] > 100_000
    assert line["allreduce_ms"] >= 0 and len(rk["allreduce_ms_mean_by_rank"]) == 8
    e2e = line["value_end_to_end"]
    assert e2e["pipelined"] > 0 and e2e["hits_per_pass_per_gpu"] > 0
    assert e2e["pipelined_sustained"] > 0 and e2e["sustained_hits_check"] is True      # one stream over four passes: four times the hits


def test_bench_sweep_world_of_eight_on_one_gpu():
    """configs[4] (host-streamed sweep) with the spans sharded over eight ranks on one GPU: the ranks' window ranges tile the
    sweep, the all-reduced per-motif window counts == the sum of the ranks' own."""
    line = _bench_ranks_on_one_gpu(["--gpus", "8", "--workload", "c5", "--genome-mbp", "240", "--steps", "1", "--warmup", "1",
                                    "--min-warm-seconds", "0"])
    assert line["n_gpus"] == 8 and line["value"] > 0
    cc = line["counts_check"]
    assert cc["allreduce_equals_sum_of_rank_counts"] is True and cc["window_ranges_tile_the_sweep"] is True and cc["ranks_with_spans"] == 8
    assert line["modes"]["hits_packed"]["sites_per_step_per_gpu"] > 0


def test_bench_collectives_over_rccl_on_a_one_rank_communicator():
    """The only RCCL run a one-GPU box allows: MS_BENCH_FORCE_PG=1 makes a single rank open the process group (backend "nccl" = RCCL,
    bound to its device) and take every N > 1 branch of bench.py -- the int64 all-reduce on the vector the library hands over, the
    all_gathers of the rank report and of the end-to-end block, the barriers, the per-rank oracle sample.  What is NOT covered: more
    than one device (xGMI, one rank per GPU); the gloo tests above cover the sharding across ranks."""
    env = dict(os.environ, MS_BENCH_FORCE_PG="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MS_BENCH_SHARE_GPU", "MS_BENCH_BACKEND", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--regions-per-set", "40000", "--steps", "3", "--warmup", "1", "--min-warm-seconds", "0",
                          "--no-cpu-baseline", "--no-api", "--no-scale-projection"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["rccl"]["backend"] == "nccl" and line["rccl"]["world"] == 1 and line["rccl"]["one_rank_per_device"] is True and line["rccl"]["nccl_version"]
    assert line["counts_check"]["allreduce_equals_sum_of_rank_counts"] is True and line["counts_check"]["max_regions_with_site"] > 0
    rk = line["ranks"]
    assert rk["shards_tile_every_set"] is True and rk["parity_sample"]["ranks_identical_to_oracle"] == 1 and rk["parity_sample"]["hits_checked"] > 10_000
    assert line["allreduce_ms"] > 0                                  # a device-side collective really ran inside the timed steps
    pr = line["value_end_to_end"]["per_rank"]
    assert pr is not None and len(pr["ms_per_pass_pipelined"]) == 1 and pr["ms_per_pass_pipelined"][0] > 0


def test_bench_refuses_more_ranks_than_gpus():
    """--gpus N with N > visible devices: one line, non-zero exit, nothing spawned (and nothing generated)."""
    import time
    n = _lib.device_count() + 3
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MS_BENCH_SHARE_GPU"):
        env.pop(k, None)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "tiny"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0 and "nothing was started" in out.stderr and not out.stdout.strip()
    assert time.time() - t0 < 120


# ----------------------------------------------------------- scan-once for overlapping regions --

@pytest.mark.parametrize("strand,dedup", [(3, False), (1, True), (2, False)])
def test_overlapping_regions_are_scanned_once_with_identical_result(oracle, jaspar579, strand, dedup):
    """Peaks +- window/2 closer together than the window, nested / duplicate / empty / clipped regions, in shuffled order
    (cli/scan.py:43-48, 76-86): ms_scan_regions_once == the per-region scan == the oracle, while scanning only the union."""
    rng = np.random.default_rng(77)
    sel = rng.choice(579, size=150, replace=False)
    mats = synth.matrices_of(jaspar579["pwm_values"], jaspar579["widths"])
    vals = np.concatenate([mats[i].ravel() for i in sel])
    widths = jaspar579["widths"][sel]
    cutoffs = jaspar579["cutoffs"]["1e-3"][sel]
    lens = [60_000, 25_000, 900, 40_000]
    chroms = {}
    for i, L in enumerate(lens):
        b, _ = synth.make_regions(1, L, seed=90 + i, frac_n=0.0)
        b[L // 3:L // 3 + 70] = ord("N")
        chroms[f"c{i}"] = b
    ci, st, en = [], [], []
    for i, L in enumerate(lens):
        summits = np.cumsum(rng.integers(20, 300, size=max(2, L // 160)))          # mean spacing 160 << window 500: every base in ~3 regions
        summits = summits[summits < L]
        for sm in summits.tolist():
            ci.append(i); st.append(max(sm - 250, 0)); en.append(min(sm + 250, L))      # scanner.py:81-83 clipping
    for _ in range(60):                                                                 # nested, duplicate, tiny and empty regions
        i = int(rng.integers(0, len(lens)))
        a = int(rng.integers(0, lens[i]))
        ci.append(i); st.append(a); en.append(min(lens[i], a + int(rng.choice([0, 3, 17, 120, 2000]))))
    ci += ci[:15]; st += st[:15]; en += en[:15]
    perm = rng.permutation(len(ci))
    ci, st, en = np.array(ci)[perm], np.array(st)[perm], np.array(en)[perm]
    total, union = int((en - st).sum()), _lib.union_bases(ci, st, en)
    assert union < 0.45 * total
    pw = _lib.PwmSet(vals, widths, cutoffs)
    rg = _lib.ResidentGenome(chroms)
    once = _lib.scan_regions_once(pw, rg, ci, st, en, strand)
    assert once.stats()["n_bases"] <= union + 1 and once.stats()["n_bases"] >= 0.9 * union      # spans = merged overlapping regions
    sq = rg.extract(ci, st, en)
    per_region = _lib.scan(pw, sq, strand)
    assert once.stats()["n_windows"] == per_region.stats()["n_windows"]
    if dedup:
        once.dedup(pw); per_region.dedup(pw)
    got, want = once.hits(), per_region.hits()
    assert len(want["pos"]) > 20_000
    assert_same_hits(got, want)
    assert np.array_equal(once.region_counts(), per_region.region_counts())
    if not dedup:
        names = list(chroms)
        seqs = b"".join(chroms[names[c]][a:b].tobytes() for c, a, b in zip(ci, st, en))
        off = np.concatenate([[0], np.cumsum(en - st)])
        assert_same_hits(got, oracle.scan_arrays(vals, widths, cutoffs, seqs, off, strand, 8))
    # the Scanner takes this path by itself for such a list

    class Reg:
        def __init__(self, c, s, e):
            self.chrom, self.start, self.end, self.summit = c, s, e, (s + e) // 2

    class P:
        def __init__(self, m, c, w):
            self.matrix, self.cutoffs, self.length = m, {"1e-3": c}, int(w)

    names = list(chroms)
    sc = scanner.Scanner(rg, [Reg(names[c], int(a), int(b)) for c, a, b in zip(ci, st, en)], 0, {3: "both", 1: "+", 2: "-"}[strand],
                         "1e-3", remove_dup=dedup)
    assert sc._as_overlapping() is not None
    a = sc.scan_motifs_arrays([P(mats[i], c, w) for i, c, w in zip(sel, cutoffs, widths)])
    assert np.array_equal(a["region"], want["seq_idx"]) and np.array_equal(a["start"], st[want["seq_idx"]] + want["pos"])
    assert np.array_equal(a["score"], want["score"])


def test_integration_stub_of_the_cli_edits_on_the_gpu(oracle, tmp_path):
    """INTEGRATION.md section 4, executed on the GPU: (a) `scan_and_enrich` -- input sites through Scanner.scan_motifs, the control set
    through Scanner.count_regions_with_sites (no site leaves the device) -- gives the enrichment rows stats.py:18-45 computes from the
    oracle's nested lists; (b) `run_sharded` with the library's own local scan (dist.gpu_scan: the shard through an ms_stream) as one
    rank, and as the two ranks of a 2-GPU job run one after the other (no process group: each call returns its LOCAL counts, which must
    add up to the whole job's)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("dist_gloo_helpers", os.path.join(ROOT, "tests", "test_dist_gloo.py"))
    hlp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hlp)
    stub = hlp._cli_stub(str(tmp_path))
    genome, pwms, regions, control = hlp.cli_job(n_pwms=40)
    sc_in, sc_ctl = scanner.Scanner(genome, regions, window_size=200), scanner.Scanner(genome, control, window_size=200)
    nested_in = oracle.deduplicate_motif_sites(hlp.oracle_nested(pwms, sc_in), [p.length for p in pwms])
    nested_ctl = hlp.oracle_nested(pwms, sc_ctl)
    want = hlp.expected_enrichment(pwms, nested_in, nested_ctl)
    motif_sites, results = stub.scan_and_enrich(genome, pwms, regions, control, 200, "both", "1e-4")
    hlp.same_results(results, want)
    assert motif_sites == [[[tuple(s) for s in per] for per in m] for m in nested_in]        # (a)'s sites are the reference's, de-duplicated
    assert sum(w[1] for w in want) > 40
    sites1, results1 = stub.run_sharded(genome, pwms, regions, control, 200, "both", "1e-4", 0, 1)
    hlp.same_results(results1, want)
    parts = [stub.run_sharded(genome, pwms, regions, control, 200, "both", "1e-4", k, 2) for k in range(2)]
    assert [sum(x) for x in zip(*[[r.n_input for r in res] for _, res in parts])] == [w[1] for w in want]
    assert [sum(x) for x in zip(*[[r.n_control for r in res] for _, res in parts])] == [w[2] for w in want]
    assert parts[0][0]["rows"][1] == parts[1][0]["rows"][0]
    raw_in = hlp.oracle_nested(pwms, sc_in)                                                     # the sharded scan leaves de-duplication to the writer
    for m in range(len(pwms)):
        flat = [(ri, st.start, st.score, 1 if st.strand == "+" else 2) for ri, per in enumerate(raw_in[m]) for st in per]
        got = []
        for s, _ in [(sites1, None)]:
            a, b = int(s["motif_offsets"][m]), int(s["motif_offsets"][m + 1])
            got += list(zip(s["region"][a:b].tolist(), s["start"][a:b].tolist(), s["score"][a:b].tolist(), s["strand"][a:b].tolist()))
        assert got == flat
        got2 = []
        for s, _ in parts:
            a, b = int(s["motif_offsets"][m]), int(s["motif_offsets"][m + 1])
            got2 += list(zip(s["region"][a:b].tolist(), s["start"][a:b].tolist(), s["score"][a:b].tolist(), s["strand"][a:b].tolist()))
        assert got2 == flat

