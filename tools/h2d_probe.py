#!/usr/bin/env python3
"""Upload rate probe: pinned host memory -> HBM through ms_seqset_create (H2D + pack), alone on the device and beside a running scan.
Prints one JSON line.  Needs an MI355X."""
import json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motifscan_amd import _lib, synth


def main():
    n = 128 << 20
    rng = np.random.default_rng(1)
    pb = _lib.PinnedBuffer(n)
    pb.array[:] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]
    offs = np.array([0, n], dtype=np.int64)
    out = {}

    def upload(reps):
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            s = _lib.SeqSet(pb.array, offs)
            t.append(time.perf_counter() - t0)
            s.close()
        return t

    upload(2)
    t = upload(8)
    out["alone_GBps"] = n / min(t) / 1e9
    out["alone_ms"] = [round(x * 1e3, 2) for x in t]
    # beside a scan: a resident 128 Mbase scan loops on another thread
    vals, widths, cutoffs = synth.load_motif_set(579)
    pw = _lib.PwmSet(vals, widths, cutoffs)
    seqs = _lib.SeqSet(pb.array, np.arange(0, n + 1, 1 << 20, dtype=np.int64))
    stop = threading.Event()
    n_scans = [0]

    def scan_loop():
        _lib.set_device(0)
        while not stop.is_set():
            r = _lib.scan(pw, seqs, 3, 0)
            r.close()
            n_scans[0] += 1

    th = threading.Thread(target=scan_loop)
    th.start()
    time.sleep(0.5)
    t = upload(12)
    stop.set(); th.join()
    out["beside_scan_GBps_best"] = n / min(t) / 1e9
    out["beside_scan_GBps_mean"] = n / (sum(t) / len(t)) / 1e9
    out["beside_ms"] = [round(x * 1e3, 2) for x in t]
    out["scans_done"] = n_scans[0]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
