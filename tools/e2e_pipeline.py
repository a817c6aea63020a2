#!/usr/bin/env python3
"""PCIe-inclusive throughput with the three stages of a batch overlapped by three host threads:
    uploader    host ASCII -> HBM + pack          (ms_seqset_create)
    scanner     scan                              (ms_scan)
    downloader  hit arrays -> pinned host views   (ms_result_hits_host)
The library's compute streams are non-blocking and the copies run outside the per-device lock, so batch i's
copy-out, batch i+1's scan and batch i+2's upload proceed together.  Usage: python tools/e2e_pipeline.py [n_batches]"""
import os, queue, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motifscan_amd import _lib, synth

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 24
_lib.set_device(0)
wl = synth.workload("c4shard")
sets = wl["sets"]
pw = _lib.PwmSet(wl["pwm_values"], wl["widths"], wl["cutoffs"])
units_per_batch = int(sets[0][1][-1]) * wl["n_pwms"]

def serial(n):
    t0 = time.perf_counter(); hits = 0
    for i in range(n):
        b, o = sets[i % 2]
        sq = _lib.SeqSet(b, o); r = _lib.scan(pw, sq, 3); h = r.hits(copy=False); hits += len(h["pos"]); r.close(); sq.close()
    return time.perf_counter() - t0, hits

def pipelined(n):
    q_up, q_res = queue.Queue(maxsize=2), queue.Queue(maxsize=2)
    hits = [0]
    def uploader():
        _lib.set_device(0)
        for i in range(n):
            b, o = sets[i % 2]
            q_up.put(_lib.SeqSet(b, o))
        q_up.put(None)
    def scanner():
        _lib.set_device(0)
        while True:
            sq = q_up.get()
            if sq is None:
                q_res.put(None); return
            r = _lib.scan(pw, sq, 3)
            sq.close()
            q_res.put(r)
    def downloader():
        _lib.set_device(0)
        while True:
            r = q_res.get()
            if r is None:
                return
            hits[0] += len(r.hits(copy=False)["pos"])
            r.close()
    th = [threading.Thread(target=f) for f in (uploader, scanner, downloader)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return time.perf_counter() - t0, hits[0]

serial(4)                                            # warm pools and caches
for rep in range(2):
    ts, hs = serial(n_batches)
    tp, hp = pipelined(n_batches)
    assert hs == hp
    print(f"rep {rep}: {n_batches} batches of 62.5 Mbase x {wl['n_pwms']} PWMs, host ASCII in -> pinned hit arrays out: "
          f"serial {1e3 * ts / n_batches:.2f} ms/batch = {units_per_batch * n_batches / ts:.3e} U/s;  "
          f"3 host threads {1e3 * tp / n_batches:.2f} ms/batch = {units_per_batch * n_batches / tp:.3e} U/s  ({hs} hits)", flush=True)
