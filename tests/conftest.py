import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small():
    with open(os.path.join(GOLDEN, "ref_small.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def rnd():
    """Seeded random PWMs / sequences / hit lists produced by the real reference."""
    d = np.load(os.path.join(GOLDEN, "ref_random.npz"))
    out = {k: d[k] for k in d.files}
    widths = out["widths"]
    vals = out["pwm_values"]
    mats, o = [], 0
    for w in widths:
        mats.append(vals[o:o + 4 * w].reshape(4, w).copy())
        o += 4 * w
    out["mats"] = mats
    off = out["seq_offsets"]
    raw = out["seq_bytes"].tobytes()
    out["seqs"] = [raw[off[i]:off[i + 1]].decode() for i in range(len(off) - 1)]
    keys = [str(k) for k in out["cutoff_keys"]]
    out["cutoff_by_key"] = {k: out["cutoffs"][:, i].copy() for i, k in enumerate(keys)}
    return out


@pytest.fixture(scope="session")
def jaspar579():
    d = np.load(os.path.join(ROOT, "motifscan_amd", "data", "synth_jaspar579.npz"))       # package data: the benchmark motif set
    keys = [str(k) for k in d["cutoff_keys"]]
    return {"widths": d["widths"], "pwm_values": d["pwm_values"],
            "cutoffs": {k: d["cutoffs"][:, i].copy() for i, k in enumerate(keys)}, "bg": d["bg"]}


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o
