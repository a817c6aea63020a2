#!/usr/bin/env python3
"""
Simulation for the 3-slot ("reference base") operand encoding of the pre-filter: how many candidates per reference hit would the
delta quantiser let through, against the current 4-slot grid quantiser (ms_plan.cpp, quantize_strand_f6)?
    acc4 = 56 - sum_c gridfloor(min(floor(d_c(b) s), 60))                                   (N-free windows; current)
    acc3 = 56 - sum_c dq'_c(b),  dq'_c(b) = a_c - g_c(b),  g_c(b) on the signed e2m3 grid (units of 1/8), dq' <= floor(d s)
Pure numpy, CPU.  Usage: python tools/delta_quant_sim.py [--windows N] [--p 1e-4] [--set path.npz]
"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GRID = np.array(sorted(set(list(range(0, 16)) + list(range(16, 32, 2)) + list(range(32, 61, 4)))))
SGRID = np.array(sorted(set((-GRID).tolist() + GRID.tolist())))


def grid_floor(q):
    q = np.asarray(q)
    return np.where(q <= 16, q, np.where(q <= 32, q & ~1, q & ~3))


def strand_tables(e, T, levels=56, clamp=60):
    """e [4, W] effective matrix.  Returns (dq4 [4,W], dq3 [4,W]) integer deficits, or None if the motif leaves the filter."""
    hi = e.max(axis=0)
    budget = hi.sum() - T
    if not budget > 0:
        return None
    s = (levels + 0.5) / budget
    q = np.floor(np.minimum((hi - e) * s * (1 - 1e-12) - 1e-7, 1e6))
    q = np.maximum(q, 0).astype(np.int64)
    qc = np.minimum(q, clamp)
    dq4 = grid_floor(qc)
    # delta form: choose a = dq'(A) in [lo .. qc[0]] so that the other three land on the grid with the least weighted loss
    W = e.shape[1]
    dq3 = np.zeros_like(qc)
    for c in range(W):
        best = None
        for a in range(0, int(qc[0, c]) + 1):
            vals = [a]
            loss = (qc[0, c] - a) * wloss(qc[0, c])
            for b in (1, 2, 3):
                # largest dq' = a - g <= qc[b,c] with g on the signed grid  <=>  smallest g >= a - qc[b,c]
                need = a - int(qc[b, c])
                g = SGRID[np.searchsorted(SGRID, need)] if need <= SGRID[-1] else None
                if g is None:
                    vals = None
                    break
                v = a - int(g)
                vals.append(v)
                loss += (qc[b, c] - v) * wloss(qc[b, c])
            if vals is None:
                continue
            if best is None or loss < best[0] - 1e-12:
                best = (loss, vals)
        dq3[:, c] = best[1]
    return dq4, dq3


def wloss(q):
    """weight of losing one level on a base whose true deficit is q levels: near-best bases are what near-hit windows are made of"""
    return 1.0 / (1.0 + q / 4.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=200_000)
    ap.add_argument("--p", default="1e-4")
    ap.add_argument("--set", default=os.path.join(ROOT, "motifscan_amd", "data", "synth_jaspar579.npz"))
    ap.add_argument("--max-motifs", type=int, default=579)
    a = ap.parse_args()
    d = np.load(a.set)
    keys = [str(k) for k in d["cutoff_keys"]]
    cut = d["cutoffs"][:, keys.index(a.p)]
    widths, vals, bg = d["widths"], d["pwm_values"], d["bg"]
    rng = np.random.default_rng(5)
    L = a.windows + 64
    seq = rng.choice(4, size=L, p=bg).astype(np.int64)
    tot = {"hits": 0, "c4": 0, "c3": 0, "loss_levels": 0.0, "cols": 0, "neg": 0}
    per_motif = []
    o = 0
    for p, W in enumerate(widths[:a.max_motifs]):
        W = int(W)
        m = vals[o:o + 4 * W].reshape(4, W)
        o += 4 * W
        max_raw = np.maximum(m.max(axis=0), 0).sum()
        absmax = np.abs(m).max(axis=0).sum()
        T = (cut[p] - 1e-10) * max_raw - 1e-9 * (1 + absmax)
        idx = np.arange(a.windows)[:, None] + np.arange(W)[None, :]
        codes = seq[idx]                                   # [n, W]
        cols = np.arange(W)
        h = c4 = c3 = 0
        for sd in (0, 1):
            e = m if sd == 0 else m[::-1, ::-1]
            tabs = strand_tables(e, T)
            x = e[codes, cols].sum(axis=1)
            hit = x >= T
            h += int(hit.sum())
            if tabs is None:
                c4 += int(hit.sum()); c3 += int(hit.sum())
                continue
            dq4, dq3 = tabs
            a4 = 56 - dq4[codes, cols].sum(axis=1)
            a3 = 56 - dq3[codes, cols].sum(axis=1)
            assert not (hit & (a4 < 0)).any() and not (hit & (a3 < 0)).any(), "a quantiser lost a hit"
            c4 += int((a4 >= 0).sum()); c3 += int((a3 >= 0).sum())
            tot["cols"] += W
            tot["neg"] += int((dq3 < 0).sum())
        tot["hits"] += h; tot["c4"] += c4; tot["c3"] += c3
        per_motif.append((W, h, c4, c3))
    print(f"p={a.p} windows={a.windows} motifs={len(per_motif)}: hits {tot['hits']}  candidates 4-slot {tot['c4']} ({tot['c4'] / max(tot['hits'], 1):.3f}/hit)  "
          f"3-slot {tot['c3']} ({tot['c3'] / max(tot['hits'], 1):.3f}/hit)  ratio {tot['c3'] / max(tot['c4'], 1):.3f}  negative dq' entries {tot['neg']}")
    pm = np.array(per_motif)
    for lo, hi_ in ((5, 7), (8, 10), (11, 15), (16, 20), (21, 30)):
        k = (pm[:, 0] >= lo) & (pm[:, 0] <= hi_)
        if k.any():
            print(f"  W {lo:2d}..{hi_:2d}: {int(k.sum()):3d} motifs  hits {pm[k, 1].sum():8d}  4-slot {pm[k, 2].sum() / max(pm[k, 1].sum(), 1):.3f}/hit  3-slot {pm[k, 3].sum() / max(pm[k, 1].sum(), 1):.3f}/hit")


if __name__ == "__main__":
    main()
