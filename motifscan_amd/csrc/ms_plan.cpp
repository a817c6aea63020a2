// ms_plan.cpp -- host side of the pre-filter: quantise every PWM into fp6 operand rows that can NEVER miss a window the
// reference would report (with or without non-ACGT bases in it), and cut the rows into LDS tiles.  Layout: ms_internal.h.
//
// Reference arithmetic being bounded (cscore.c:340-390): with NS the set of the window's non-ACGT columns (they add nothing, :345-353)
//     s   = fl64( sum_{c not in NS} M[code_c][c] )            (column order)
//     hit = fl64( fl64(s / max_raw) - cutoff ) >= -1e-10
// With x the exact real sum of the same doubles, a reported hit implies
//     x >= T := (cutoff - 1e-10) * max_raw - E,   E = 1e-9 * (1 + sum_c max_b |M[b][c]|)
// (E is ~10^3 times the worst-case fp64 rounding of the W adds, the divide and the subtract).
//
// Per strand, with e[b][c] the effective matrix (cscore.c:351), hi_c = max_b e[b][c] and the deficit d_c(b) = hi_c - e[b][c] >= 0:
//     x = sum_{c not in NS} hi_c - sum_{c not in NS} d_c(code_c),   budget := sum_c hi_c - T,   s := (56 + 1/2) / budget.
// Deficits are quantised DOWN: dq_c(b) = grid_floor(min(floor(d_c(b) * s), 60)) <= d_c(b) * s (60 sinks a window alone).
// Offsets: t_c = the largest multiple of 4 that is <= min(16, hi_c * s) and >= 0; a column whose best base is negative (it can
// only occur in arbitrary matrices, never in log-odds of normalised probabilities) gets t_c = 0 and its shortfall pen_c =
// ceil(-hi_c * s) is added to the bias.  Bias b0 = 56 + pen - sum_c t_c (pen = sum pen_c rounded up to a multiple of 4), kept
// inside the e2m3 range by lowering offsets if it falls below -60.  The kernel evaluates, in units of 1/8,
//     acc = b0 + sum_{c not in NS} (t_c - dq_c(code_c)).
// Claim: a reported hit has acc >= 0.  Proof: hit => sum_{c not in NS} d_c <= sum_{c not in NS} hi_c - T, so
//     sum_{c not in NS} dq_c <= s * (sum_{c not in NS} hi_c - T) = s * budget - s * sum_{c in NS} hi_c = 56.5 - s * sum_{c in NS} hi_c, and
//     acc >= 56 + pen - sum_{c in NS} t_c - 56.5 + s * sum_{c in NS} hi_c = -1/2 + pen - sum_{c in NS} (t_c - s * hi_c)
//         >= -1/2 + pen - sum_{c in NS, hi_c < 0} (-s * hi_c) >= -1/2          (t_c <= s * hi_c wherever hi_c >= 0; pen covers the rest),
// and acc is an integer.  For NS empty acc = 56 - sum dq: the filter is as tight as the grid allows; with NS = every column
// acc = b0, negative whenever the motif's columns can carry more than the budget -- an all-N window is then not even a candidate.
// Whatever passes is re-scored in fp64 in the reference's order, so the pre-filter decides nothing by itself; it only must not
// lose hits (proved on the CPU from the physical operand image: tests/test_host_cabi.py).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "ms_internal.h"

namespace ms {

namespace {

// Can the motif take the pre-filter at all, and at which threshold T on the exact real sum?
bool filter_threshold(const double *m, int W, double cutoff, double max_raw, double *T) {
    bool ok = W >= 1 && W <= kMaxFastWidth && std::isfinite(max_raw) && max_raw > 0 && std::isfinite(cutoff);
    double abs_sum = 0;
    for (int c = 0; ok && c < W; c++) {
        double colmax = 0;
        for (int b = 0; b < 4; b++) {
            const double v = m[(int64_t) b * W + c];
            if (!std::isfinite(v) || std::fabs(v) > 1e9) { ok = false; break; }
            colmax = std::max(colmax, std::fabs(v));
        }
        abs_sum += colmax;
    }
    if (!ok) return false;
    const double E = 1e-9 * (1.0 + abs_sum);
    *T = (cutoff - 1e-10) * max_raw - E;
    return true;
}

struct F6Strand {
    int8_t u[kMaxFastWidth][4];     // t_c - dq_c(b), units of 1/8
    int8_t bias;                    // b0
};

inline int f6_grid_floor(int q) { return q <= 16 ? q : (q <= 32 ? (q & ~1) : (q & ~3)); }

void f6_dead(F6Strand *out) {                   // acc = -1/8 for every window: never a candidate
    std::memset(out->u, 0, sizeof(out->u));
    out->bias = -1;
}

// Returns false if the strand needs the fp64 path (a threshold so low that filtering is pointless, or out of the grid's range).
// Bq budget levels, deficits of `clamp` (> Bq, on the grid) and more stored as `clamp`: 56 / 60 wherever the row can carry it; 36 / 40
// for the paired rows of motifs of 16 ... 23 columns, whose field must stay inside +-1024 (24 x 40 + 60 < 1024).
bool quantize_strand_f6(const double e[4][kMaxFastWidth], int W, double T, int Bq, int clamp, F6Strand *out, bool *alln_can_hit) {
    f6_dead(out);
    double hi[kMaxFastWidth], sum_hi = 0, sum_hi_pos = 0, lowest = 0;
    for (int c = 0; c < W; c++) {
        hi[c] = std::max(std::max(e[0][c], e[1][c]), std::max(e[2][c], e[3][c]));
        sum_hi += hi[c];
        sum_hi_pos += std::max(hi[c], 0.0);                                        // a non-ACGT base adds 0: better than a negative best base
        lowest += std::min(std::min(std::min(e[0][c], e[1][c]), std::min(e[2][c], e[3][c])), 0.0);
    }
    if (!(sum_hi_pos >= T)) return true;        // dead: no window, with or without non-ACGT bases, reaches T
    const double budget = sum_hi - T;
    if (!(budget >= 0)) return false;           // only windows with non-ACGT bases could reach T: not worth a table
    if (!(T > lowest)) return false;            // every window passes
    if (!(T > 0)) *alln_can_hit = true;         // a window of non-ACGT bases only scores 0 and may be reported
    const double s = budget > 0 ? ((double) Bq + 0.5) / budget : 1e300;
    int t[kMaxFastWidth], dq[kMaxFastWidth][4];
    long pen = 0, sum_t = 0;
    for (int c = 0; c < W; c++) {
        for (int b = 0; b < 4; b++) {
            const double d = hi[c] - e[b][c];
            double q = d <= 0 ? 0.0 : std::floor(std::min(d * s * (1 - 1e-12) - 1e-7, 1e6));
            if (!(q >= 0)) q = 0;
            dq[c][b] = f6_grid_floor((int) std::min(q, (double) clamp));  // always DOWN (to the grid, to the range): `clamp` sinks the window alone
        }
        if (hi[c] >= 0) {
            double lim = std::floor(std::min(hi[c] * s * (1 - 1e-12) - 1e-7, 1e6));
            if (!(lim >= 0)) lim = 0;
            t[c] = std::min(16, 4 * ((int) lim / 4));
        } else {
            t[c] = 0;
            const double p = std::ceil(std::min(-hi[c] * s * (1 + 1e-12) + 1e-7, 1e6));
            pen += (long) p;
        }
        sum_t += t[c];
    }
    pen = (pen + 3) / 4 * 4;
    long b0 = Bq + pen - sum_t;
    for (int c = W - 1; b0 < -60 && c >= 0; c--)                        // keep the bias inside the grid: give offsets back
        while (b0 < -60 && t[c] > 0) { t[c] -= 4; b0 += 4; }
    if (b0 > 60 || b0 < -60 || !f6_representable((int) b0)) return false;
    for (int c = 0; c < W; c++)
        for (int b = 0; b < 4; b++) {
            const int u = t[c] - dq[c][b];
            if (!f6_representable(u)) return false;                       // (cannot happen: offsets are multiples of 4)
            out->u[c][b] = (int8_t) u;
        }
    out->bias = (int8_t) b0;
    return true;
}

// A strand as a DELTA row (ms_internal.h): for windows without non-ACGT bases acc = 56 - sum_c dq'_c(code_c) with
// dq'_c(A) = a_c and dq'_c(b) = a_c - delta_c(b), every delta on the signed e2m3 grid and every dq' at or below the true scaled
// deficit floor(d_c(b) s) (so a reported hit still has sum dq' <= 56.5, acc >= 0: the proof of the one-hot rows with NS empty).
struct D3Strand {
    int8_t delta[kMaxFastWidth][3]; // what C, G, T add on top of A, units of 1/8
    int32_t a_sum;                  // sum_c dq'_c(A): the bias is 56 - a_sum
    bool dead;                      // never a candidate (strand not asked for, or no window reaches the threshold)
};

inline int grid_ceil_signed(int need) {         // the smallest value of the signed e2m3 grid that is >= need (|need| <= 60)
    for (int g = need; g <= 60; g++)
        if (f6_representable(g)) return g;
    return 60;
}

// false: the strand cannot ride a delta row (only ever for strands quantize_strand_f6 refuses as well)
bool quantize_strand_d3(const double e[4][kMaxFastWidth], int W, double T, int levels, D3Strand *out) {
    std::memset(out->delta, 0, sizeof(out->delta));
    out->a_sum = 0;
    out->dead = true;
    double hi[kMaxFastWidth], sum_hi = 0, sum_hi_pos = 0, lowest = 0;
    for (int c = 0; c < W; c++) {
        hi[c] = std::max(std::max(e[0][c], e[1][c]), std::max(e[2][c], e[3][c]));
        sum_hi += hi[c];
        sum_hi_pos += std::max(hi[c], 0.0);
        lowest += std::min(std::min(std::min(e[0][c], e[1][c]), std::min(e[2][c], e[3][c])), 0.0);
    }
    if (!(sum_hi_pos >= T)) return true;        // dead, as in the one-hot form
    const double budget = sum_hi - T;
    if (!(budget >= 0) || !(T > lowest)) return false;
    const double s = budget > 0 ? ((double) levels + 0.5) / budget : 1e300;
    const int clamp = std::min(60, kDeltaMaxSum / std::max(W, 1));      // W x clamp <= kDeltaMaxSum: the field never leaves its 11 bits
    if (clamp <= levels / 2) return false;
    for (int c = 0; c < W; c++) {
        int qc[4];
        for (int b = 0; b < 4; b++) {
            const double d = hi[c] - e[b][c];
            double q = d <= 0 ? 0.0 : std::floor(std::min(d * s * (1 - 1e-12) - 1e-7, 1e6));
            if (!(q >= 0)) q = 0;
            qc[b] = (int) std::min(q, (double) clamp);
        }
        // a = dq'(A) in [0, qc[A]]: the choice that loses the least against the true deficits, a level lost on a near-best base
        // (what near-hit windows are made of) counting more than one on a base that sinks the window anyway
        auto weight = [](int q) { return 1.0 / (1.0 + q / 4.0); };
        double best_loss = -1;
        int best_a = 0, best_g[3] = {0, 0, 0};
        for (int a = qc[0]; a >= 0; a--) {
            double loss = (qc[0] - a) * weight(qc[0]);
            int g[3];
            for (int b = 1; b < 4; b++) {
                g[b - 1] = grid_ceil_signed(a - qc[b]);                 // dq'(b) = a - g <= qc[b], as large as the grid allows
                loss += (qc[b] - (a - g[b - 1])) * weight(qc[b]);
            }
            if (best_loss < 0 || loss < best_loss - 1e-12) { best_loss = loss; best_a = a; best_g[0] = g[0]; best_g[1] = g[1]; best_g[2] = g[2]; }
        }
        out->a_sum += best_a;
        for (int b = 0; b < 3; b++) out->delta[c][b] = (int8_t) best_g[b];
    }
    out->dead = false;
    return true;
}

struct FastMotif {
    int32_t id;
    int32_t W;
    F6Strand f6[2];                 // 56 levels
    F6Strand f6w[2];                // 36 levels (motifs of 16 ... kPairWideMaxWidth columns that may ride a paired row)
    D3Strand d3[2];                 // the same strands as delta rows (motifs of <= kDeltaMaxWidth columns)
    bool wide_ok;
    bool use_wide;
    bool d3_ok;
};

}  // namespace

int build_plan(const double *values, const int64_t *val_off, const int32_t *widths, const double *cutoffs,
               const double *max_raw, int32_t n_pwms, int strand_mask, size_t lds_budget, bool pair_rows, PrefilterPlan *plan,
               bool delta_rows) {
    *plan = PrefilterPlan();
    plan->strand_mask = strand_mask;
    std::vector<FastMotif> fast;
    fast.reserve(n_pwms);
    for (int32_t p = 0; p < n_pwms; p++) {
        const int W = widths[p];
        const double *m = values + val_off[p];
        double T = 0;
        bool ok = filter_threshold(m, W, cutoffs[p], max_raw[p], &T);
        FastMotif fm;
        fm.id = p;
        fm.W = W;
        bool alln = false;
        for (int sd = 0; sd < 2; sd++) { f6_dead(&fm.f6[sd]); f6_dead(&fm.f6w[sd]); }     // never a candidate unless quantised below
        fm.wide_ok = pair_rows && W > kPairMaxWidth && W <= kPairWideMaxWidth;
        fm.use_wide = false;
        fm.d3_ok = W <= kDeltaMaxWidth;
        for (int sd = 0; sd < 2; sd++) { std::memset(fm.d3[sd].delta, 0, sizeof(fm.d3[sd].delta)); fm.d3[sd].a_sum = 0; fm.d3[sd].dead = true; }
        for (int sd = 0; ok && sd < 2; sd++) {
            if (!(strand_mask & (1 << sd))) continue;                    // strand not asked for
            double e[4][kMaxFastWidth];
            for (int b = 0; b < 4; b++)
                for (int c = 0; c < W; c++)
                    e[b][c] = sd == 0 ? m[(int64_t) b * W + c] : m[(int64_t) (3 - b) * W + (W - 1 - c)];   // cscore.c:351
            ok = quantize_strand_f6(e, W, T, kF6Levels, 60, &fm.f6[sd], &alln);
            if (ok && fm.wide_ok) { bool alln2 = false; fm.wide_ok = quantize_strand_f6(e, W, T, kPairWideLevels, 40, &fm.f6w[sd], &alln2); }
            if (ok && fm.d3_ok) fm.d3_ok = quantize_strand_d3(e, W, T, kF6Levels, &fm.d3[sd]);
        }
        if (ok) { fast.push_back(fm); plan->alln_can_hit = plan->alln_can_hit || alln; }
        else plan->exact_motifs.push_back(p);
    }
    // Row tiles, in this order: delta rows, plain rows of > 20 columns, paired rows, plain rows of 16 ... 20 columns (narrow to wide within each).  A row tile serves passes WITH
    // non-ACGT bases (kFamilyN), passes without (kFamilyClean) or both; every motif is covered exactly once in either kind of pass:
    //   with delta rows:    N passes     = paired rows (<= 15 columns) + plain rows (16 ... 20 columns) + plain rows (wider)
    //                       clean passes = delta rows (<= 20 columns)                                    + plain rows (wider)
    //   without:            every pass   = paired rows + plain rows.
    // Motifs of 16 ... kPairWideMaxWidth columns may go either way in the one-hot family (three half-blocks at 36 levels for 32 motifs,
    // or two k-blocks at 56 levels for 16): the narrowest n_w of them ride paired rows, n_w chosen for the least instructions.
    const bool both = strand_mask == 3;
    const int sd_single = strand_mask == 2 ? 1 : 0;
    std::stable_sort(fast.begin(), fast.end(), [](const FastMotif &x, const FastMotif &y) { return x.W < y.W; });
    bool use_delta = delta_rows && pair_rows;
    for (const FastMotif &fm : fast)
        if (fm.W <= kDeltaMaxWidth && !fm.d3_ok) use_delta = false;         // (cannot happen for a strand the one-hot form accepts)
    plan->delta = use_delta;
    struct RowTile { int kind; int nk; int family; std::vector<int> members; size_t off; };     // kind: 0 plain, 1 paired, 2 delta; members: indices into `fast`
    auto nk_of = [&](int i, int kind) { return kind == 2 ? delta_kb_of_width(fast[i].W) : (kind == 1 ? pair_kb_of_width(fast[i].W) : f6_kb_of_width(fast[i].W)); };
    // Row tiles = runs of consecutive motifs of `v` (sorted by width: a tile pays for its widest), cut so that the instruction count
    // is minimal -- e.g. the last few narrow motifs get a short tile of their own rather than riding a tile of wider ones
    auto cut = [&](const std::vector<int> &v, int kind, int family, std::vector<RowTile> *out) -> long {
        const size_t per_rt = (both ? 16 : 32) * (kind ? 2 : 1), n = v.size();
        std::vector<long> cost(n + 1, 0);
        std::vector<size_t> from(n + 1, 0);
        for (size_t i = 1; i <= n; i++) {
            cost[i] = -1;
            for (size_t j = i > per_rt ? i - per_rt : 0; j < i; j++) {
                const long cst = cost[j] + 64L * nk_of(v[i - 1], kind) + 1;         // instructions first, then the number of tiles
                if (cost[i] < 0 || cst < cost[i]) { cost[i] = cst; from[i] = j; }
            }
        }
        if (out) {
            std::vector<size_t> ends;
            for (size_t i = n; i > 0; i = from[i]) ends.push_back(i);
            size_t j = 0;
            for (size_t k = ends.size(); k-- > 0;) {
                out->push_back(RowTile{kind, nk_of(v[ends[k] - 1], kind), family, std::vector<int>(v.begin() + (long) j, v.begin() + (long) ends[k]), 0});
                j = ends[k];
            }
        }
        return cost[n];
    };
    std::vector<int> narrow, mid, rest;                                         // <= 15 columns | either way | plain only  (indices, by width)
    for (int i = 0; i < (int) fast.size(); i++) (pair_rows && fast[i].W <= kPairMaxWidth ? narrow : (fast[i].wide_ok ? mid : rest)).push_back(i);
    size_t best_w = 0;
    {
        long best = -1;
        for (size_t n_w = 0; n_w <= mid.size(); n_w++) {
            std::vector<int> pv(narrow), lv(mid.begin() + (long) n_w, mid.end());
            pv.insert(pv.end(), mid.begin(), mid.begin() + (long) n_w);
            lv.insert(lv.end(), rest.begin(), rest.end());
            std::stable_sort(lv.begin(), lv.end(), [&](int x, int y) { return fast[x].W < fast[y].W; });
            const long cst = cut(pv, 1, 0, nullptr) + cut(lv, 0, 0, nullptr);
            if (best < 0 || cst < best) { best = cst; best_w = n_w; }
        }
    }
    std::vector<RowTile> rts;
    {
        std::vector<int> pv(narrow), lv(mid.begin() + (long) best_w, mid.end()), lvA, lvB, dv;
        for (size_t i = 0; i < best_w; i++) { pv.push_back(mid[i]); fast[mid[i]].use_wide = true; }
        lv.insert(lv.end(), rest.begin(), rest.end());
        std::stable_sort(lv.begin(), lv.end(), [&](int x, int y) { return fast[x].W < fast[y].W; });
        for (int i : lv) (use_delta && fast[i].W <= kDeltaMaxWidth ? lvA : lvB).push_back(i);
        if (use_delta)
            for (int i = 0; i < (int) fast.size(); i++)
                if (fast[i].W <= kDeltaMaxWidth) dv.push_back(i);
        // the rows of the clean passes FIRST: the kernel that runs those passes loads only that prefix of an LDS tile (TileDesc::clean_len16)
        (void) cut(dv, 2, kFamilyClean, &rts);
        (void) cut(lvB, 0, kFamilyN | kFamilyClean, &rts);
        (void) cut(pv, 1, use_delta ? kFamilyN : (kFamilyN | kFamilyClean), &rts);
        (void) cut(lvA, 0, kFamilyN, &rts);
    }
    const size_t n_rt = rts.size();
    size_t total = 0;
    for (RowTile &rt : rts) {
        rt.off = total;
        total += (size_t) rt.nk * kF6BytesPerKb;
        if (rt.family & kFamilyClean) {
            plan->kb_total += rt.nk;
            plan->lds_bytes_per_position += (int64_t) rt.nk * (int64_t) kF6BytesPerKb / 64;    // A-operand bytes per window start (2 x 32 windows share a read)
        }
    }
    std::vector<uint8_t> bytes(total, 0);
    for (size_t t = 0; t < n_rt; t++) {
        const RowTile &rt = rts[t];
        uint8_t *tab = bytes.data() + rt.off;
        const int cols_per_kb = rt.kind == 2 ? kDeltaCols : (rt.kind == 1 ? kPairCols : kF6Cols);
        const int n_cols = cols_per_kb * rt.nk;
        const int n_groups = rt.kind ? 4 : 2;
        const size_t per_group = both ? 8 : 16;
        for (int gi = 0; gi < n_groups; gi++) {
            const int h = rt.kind ? gi >> 1 : gi, sel = rt.kind ? gi & 1 : 0;
            const size_t grp = plan->group_kb.size();
            plan->group_kb.push_back(rt.nk);
            plan->group_cols.push_back(n_cols);
            plan->group_info.push_back(GroupInfo{(uint32_t) rt.off, (int8_t) rt.nk, (int8_t) rt.kind, (int8_t) h, (int8_t) sel});
            plan->group_fields.resize((grp + 1) * kGroupFields, -1);
            for (int n = 0; n < kGroupFields; n++) {
                const size_t slot = per_group * (size_t) gi + (both ? (size_t) (n >> 1) : (size_t) n);
                const FastMotif *fm = slot < rt.members.size() ? &fast[rt.members[slot]] : nullptr;
                const int sd = both ? (n & 1) : sd_single;
                const int row = mfma_row_of(h, n);
                if (fm) plan->group_fields[grp * kGroupFields + n] = fm->id;
                if (rt.kind == 2) {
                    // delta row: (C, G, T) of column c in k-slots delta_slot(c % 10, b) of half-block c / 10, k-half `sel`; the bias in the
                    // row's four bias slots (delta_bias_slots)
                    const D3Strand *ds = fm && !fm->d3[sd].dead ? &fm->d3[sd] : nullptr;
                    if (ds)
                        for (int c = 0; c < fm->W; c++)
                            for (int b = 1; b < 4; b++) {
                                const int sl = delta_slot(c % kDeltaCols, b);
                                f6_put(tab, rt.nk, c / kDeltaCols, row, 8 * sel + sl / 4, sl % 4, f6_code(ds->delta[c][b - 1]));
                            }
                    const int tot = kPairOffset + (ds ? kF6Levels - ds->a_sum : -1);          // empty / dead field: acc = -1/8, never a candidate
                    int u4[4] = {0, 0, 0, 0};
                    if (!pair_bias_entries(tot, u4)) {
                        set_error("internal: no delta bias entries for %d", tot);
                        return MS_ERR_RUNTIME;
                    }
                    const DeltaBiasSlot *bs = delta_bias_slots(rt.nk);                       // (k-slot s = column s / 4, base s % 4 of the k-half's 8 x 4 grid)
                    for (int k = 0; k < 4; k++) f6_put(tab, rt.nk, bs[k].kb, row, 8 * sel + bs[k].slot / 4, bs[k].slot % 4, f6_code(u4[bs[k].u]));
                    continue;
                }
                const F6Strand *fs = fm ? (fm->use_wide ? &fm->f6w[sd] : &fm->f6[sd]) : nullptr;
                // empty field: bias -1/8 and nothing else -> never a candidate; columns past W stay +0; the bias sits in the field's
                // LAST column for all four bases (the kernel never clears that column for non-ACGT bases)
                int pb[4] = {0, 0, 0, 0};                                   // paired rows: the bias column also carries the field's offset
                if (rt.kind == 1 && !pair_bias_entries((fs ? fs->bias : -1) + kPairOffset, pb)) {
                    set_error("internal: no bias entries for %d", (fs ? fs->bias : -1) + kPairOffset);
                    return MS_ERR_RUNTIME;
                }
                for (int c = 0; c < n_cols; c++)
                    for (int b = 0; b < 4; b++) {
                        int u = 0;
                        if (c == n_cols - 1) u = rt.kind == 1 ? pb[b] : (fs ? fs->bias : -1);
                        else if (fs && c < fm->W) u = fs->u[c][b];
                        // plain: column c of k-block c / 16; paired: column c % 8 of half-block c / 8, in k-half `sel`
                        if (rt.kind == 1) f6_put(tab, rt.nk, c / kPairCols, row, kPairCols * sel + c % kPairCols, b, f6_code(u));
                        else f6_put(tab, rt.nk, c / kF6Cols, row, c % kF6Cols, b, f6_code(u));
                    }
            }
        }
    }
    for (const FastMotif &fm : fast) plan->fast_motifs.push_back(fm.id);
    plan->tables.resize(bytes.size() / 4);
    if (!bytes.empty()) std::memcpy(plan->tables.data(), bytes.data(), bytes.size());

    // LDS tiles (whole row tiles; work ~ bytes), classes = runs of equal (paired, instruction count)
    if (n_rt > 0) {
        const size_t budget = std::max<size_t>(lds_budget, (size_t) kF6MaxKb * kF6BytesPerKb);
        const size_t n_tiles = (total + budget - 1) / budget;
        const size_t target = (total + n_tiles - 1) / n_tiles;
        size_t q = 0;
        int32_t group = 0;
        while (q < n_rt) {
            TileDesc t;
            std::memset(&t, 0, sizeof(t));
            t.table_off16 = (uint32_t) (rts[q].off / 16);
            t.first_group = group;
            size_t used = 0;
            while (q < n_rt) {
                const size_t need = (size_t) rts[q].nk * kF6BytesPerKb;
                if (used > 0 && (used + need > budget || used >= target)) break;
                const bool new_class = t.n_classes == 0 || t.cls[t.n_classes - 1].nk != rts[q].nk || t.cls[t.n_classes - 1].paired != rts[q].kind ||
                                       t.cls[t.n_classes - 1].family != rts[q].family;
                if (new_class) {
                    if (t.n_classes == kMaxClasses) break;                   // (cannot happen: the row tiles are sorted by class)
                    ClassDesc &cd = t.cls[t.n_classes++];
                    cd.nk = rts[q].nk;
                    cd.n_row_tiles = 0;
                    cd.base16 = (uint32_t) (used / 16);
                    cd.first_group = group;
                    cd.paired = rts[q].kind;
                    cd.family = rts[q].family;
                }
                t.cls[t.n_classes - 1].n_row_tiles++;
                if (!rts[q].kind) t.max_nk = std::max(t.max_nk, rts[q].nk);
                used += need;
                if (rts[q].family & kFamilyClean) t.clean_len16 = (uint32_t) (used / 16);      // (the clean rows of a tile are a prefix of it: the order above)
                group += rts[q].kind ? 4 : 2;
                q++;
            }
            t.table_len16 = (uint32_t) (used / 16);
            plan->tiles.push_back(t);
        }
    }
    return MS_OK;
}

}  // namespace ms
