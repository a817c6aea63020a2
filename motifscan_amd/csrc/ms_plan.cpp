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

struct FastMotif {
    int32_t id;
    int32_t W;
    F6Strand f6[2];                 // 56 levels
    F6Strand f6w[2];                // 36 levels (motifs of 16 ... kPairWideMaxWidth columns that may ride a paired row)
    bool wide_ok;
    bool use_wide;
};

}  // namespace

int build_plan(const double *values, const int64_t *val_off, const int32_t *widths, const double *cutoffs,
               const double *max_raw, int32_t n_pwms, int strand_mask, size_t lds_budget, bool pair_rows, PrefilterPlan *plan) {
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
        for (int sd = 0; ok && sd < 2; sd++) {
            if (!(strand_mask & (1 << sd))) continue;                    // strand not asked for
            double e[4][kMaxFastWidth];
            for (int b = 0; b < 4; b++)
                for (int c = 0; c < W; c++)
                    e[b][c] = sd == 0 ? m[(int64_t) b * W + c] : m[(int64_t) (3 - b) * W + (W - 1 - c)];   // cscore.c:351
            ok = quantize_strand_f6(e, W, T, kF6Levels, 60, &fm.f6[sd], &alln);
            if (ok && fm.wide_ok) { bool alln2 = false; fm.wide_ok = quantize_strand_f6(e, W, T, kPairWideLevels, 40, &fm.f6w[sd], &alln2); }
        }
        if (ok) { fast.push_back(fm); plan->alln_can_hit = plan->alln_can_hit || alln; }
        else plan->exact_motifs.push_back(p);
    }
    // Paired rows first (narrow to wide), then plain rows (narrow to wide).  Motifs of 16 ... kPairWideMaxWidth columns may go either way
    // (three half-blocks at 36 levels for 32 motifs, or two k-blocks at 56 levels for 16): the narrowest n_w of them ride paired rows,
    // n_w chosen so that the plan's instruction count is minimal.
    const bool both = strand_mask == 3;
    const int sd_single = strand_mask == 2 ? 1 : 0;
    struct RowTile { bool paired; int nk; size_t first, count, off; };
    auto nk_of = [&](const FastMotif &a, bool pr) { return pr ? pair_kb_of_width(a.W) : f6_kb_of_width(a.W); };
    // Row tiles = runs of consecutive motifs of `v` (sorted by width: a tile pays for its widest), cut so that the instruction count
    // is minimal -- e.g. the last few narrow motifs get a short tile of their own rather than riding a tile of wider ones
    auto cut = [&](const std::vector<const FastMotif *> &v, bool pr, size_t base, std::vector<RowTile> *out) -> long {
        const size_t per_rt = (both ? 16 : 32) * (pr ? 2 : 1), n = v.size();
        std::vector<long> cost(n + 1, 0);
        std::vector<size_t> from(n + 1, 0);
        for (size_t i = 1; i <= n; i++) {
            cost[i] = -1;
            for (size_t j = i > per_rt ? i - per_rt : 0; j < i; j++) {
                const long cst = cost[j] + 64L * nk_of(*v[i - 1], pr) + 1;         // instructions first, then the number of tiles
                if (cost[i] < 0 || cst < cost[i]) { cost[i] = cst; from[i] = j; }
            }
        }
        if (out) {
            std::vector<size_t> ends;
            for (size_t i = n; i > 0; i = from[i]) ends.push_back(i);
            size_t j = 0;
            for (size_t k = ends.size(); k-- > 0;) {
                out->push_back(RowTile{pr, nk_of(*v[ends[k] - 1], pr), base + j, ends[k] - j, 0});
                j = ends[k];
            }
        }
        return cost[n];
    };
    std::stable_sort(fast.begin(), fast.end(), [](const FastMotif &a, const FastMotif &b) { return a.W < b.W; });
    std::vector<const FastMotif *> narrow, mid, rest;                           // <= 15 columns | either way | plain only
    for (const FastMotif &fm : fast) (pair_rows && fm.W <= kPairMaxWidth ? narrow : (fm.wide_ok ? mid : rest)).push_back(&fm);
    size_t best_w = 0;
    {
        long best = -1;
        for (size_t n_w = 0; n_w <= mid.size(); n_w++) {
            std::vector<const FastMotif *> pv(narrow), lv(mid.begin() + (long) n_w, mid.end());
            pv.insert(pv.end(), mid.begin(), mid.begin() + (long) n_w);
            lv.insert(lv.end(), rest.begin(), rest.end());
            std::stable_sort(lv.begin(), lv.end(), [](const FastMotif *a, const FastMotif *b) { return a->W < b->W; });
            const long cst = cut(pv, true, 0, nullptr) + cut(lv, false, 0, nullptr);
            if (best < 0 || cst < best) { best = cst; best_w = n_w; }
        }
    }
    std::vector<FastMotif> ordered;
    ordered.reserve(fast.size());
    for (const FastMotif *f : narrow) ordered.push_back(*f);
    for (size_t i = 0; i < best_w; i++) { ordered.push_back(*mid[i]); ordered.back().use_wide = true; }
    const size_t n_paired = ordered.size();
    {
        std::vector<const FastMotif *> lv(mid.begin() + (long) best_w, mid.end());
        lv.insert(lv.end(), rest.begin(), rest.end());
        std::stable_sort(lv.begin(), lv.end(), [](const FastMotif *a, const FastMotif *b) { return a->W < b->W; });
        for (const FastMotif *f : lv) ordered.push_back(*f);
    }
    fast.swap(ordered);
    std::vector<RowTile> rts;
    {
        std::vector<const FastMotif *> pv, lv;
        for (size_t i = 0; i < fast.size(); i++) (i < n_paired ? pv : lv).push_back(&fast[i]);
        (void) cut(pv, true, 0, &rts);
        (void) cut(lv, false, n_paired, &rts);
    }
    const size_t n_rt = rts.size();
    size_t total = 0;
    for (RowTile &rt : rts) {
        rt.off = total;
        total += (size_t) rt.nk * kF6BytesPerKb;
        plan->kb_total += rt.nk;
        plan->lds_bytes_per_position += (int64_t) rt.nk * (int64_t) kF6BytesPerKb / 64;        // A-operand bytes per window start (2 x 32 windows share a read)
    }
    std::vector<uint8_t> bytes(total, 0);
    for (size_t t = 0; t < n_rt; t++) {
        const RowTile &rt = rts[t];
        uint8_t *tab = bytes.data() + rt.off;
        const int cols_per_kb = rt.paired ? kPairCols : kF6Cols;
        const int n_cols = cols_per_kb * rt.nk;
        const int n_groups = rt.paired ? 4 : 2;
        const size_t per_group = both ? 8 : 16;
        for (int gi = 0; gi < n_groups; gi++) {
            const int h = rt.paired ? gi >> 1 : gi, sel = rt.paired ? gi & 1 : 0;
            const size_t grp = plan->group_kb.size();
            plan->group_kb.push_back(rt.nk);
            plan->group_cols.push_back(n_cols);
            plan->group_info.push_back(GroupInfo{(uint32_t) rt.off, (int8_t) rt.nk, (int8_t) rt.paired, (int8_t) h, (int8_t) sel});
            plan->group_fields.resize((grp + 1) * kGroupFields, -1);
            for (int n = 0; n < kGroupFields; n++) {
                const size_t slot = per_group * (size_t) gi + (both ? (size_t) (n >> 1) : (size_t) n);
                const size_t j = rt.first + slot;
                const int sd = both ? (n & 1) : sd_single;
                const int row = mfma_row_of(h, n);
                const F6Strand *fs = slot < rt.count ? (fast[j].use_wide ? &fast[j].f6w[sd] : &fast[j].f6[sd]) : nullptr;
                // empty field: bias -1/8 and nothing else -> never a candidate; columns past W stay +0; the bias sits in the field's
                // LAST column for all four bases (the kernel never clears that column for non-ACGT bases)
                int pb[4] = {0, 0, 0, 0};                                   // paired rows: the bias column also carries the field's offset
                if (rt.paired && !pair_bias_entries((fs ? fs->bias : -1) + kPairOffset, pb)) {
                    set_error("internal: no bias entries for %d", (fs ? fs->bias : -1) + kPairOffset);
                    return MS_ERR_RUNTIME;
                }
                for (int c = 0; c < n_cols; c++)
                    for (int b = 0; b < 4; b++) {
                        int u = 0;
                        if (c == n_cols - 1) u = rt.paired ? pb[b] : (fs ? fs->bias : -1);
                        else if (fs && c < fast[j].W) u = fs->u[c][b];
                        // plain: column c of k-block c / 16; paired: column c % 8 of half-block c / 8, in k-half `sel`
                        if (rt.paired) f6_put(tab, rt.nk, c / kPairCols, row, kPairCols * sel + c % kPairCols, b, f6_code(u));
                        else f6_put(tab, rt.nk, c / kF6Cols, row, c % kF6Cols, b, f6_code(u));
                    }
                if (fs) plan->group_fields[grp * kGroupFields + n] = fast[j].id;
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
                const bool new_class = t.n_classes == 0 || t.cls[t.n_classes - 1].nk != rts[q].nk || (t.cls[t.n_classes - 1].paired != 0) != rts[q].paired;
                if (new_class) {
                    if (t.n_classes == kMaxClasses) break;                   // (cannot happen: the row tiles are sorted by class)
                    ClassDesc &cd = t.cls[t.n_classes++];
                    cd.nk = rts[q].nk;
                    cd.n_row_tiles = 0;
                    cd.base16 = (uint32_t) (used / 16);
                    cd.first_group = group;
                    cd.paired = rts[q].paired ? 1 : 0;
                }
                t.cls[t.n_classes - 1].n_row_tiles++;
                if (!rts[q].paired) t.max_nk = std::max(t.max_nk, rts[q].nk);
                used += need;
                group += rts[q].paired ? 4 : 2;
                q++;
            }
            t.table_len16 = (uint32_t) (used / 16);
            plan->tiles.push_back(t);
        }
    }
    return MS_OK;
}

}  // namespace ms
