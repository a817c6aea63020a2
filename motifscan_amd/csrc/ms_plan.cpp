// ms_plan.cpp -- host side of the integer pre-filter: quantise every PWM into small integer tables
// that can NEVER miss a window the reference would report, and cut the tables into LDS tiles.
// Two table forms share the threshold derivation below:
//   engine 1 (default)  int8 rows of a matrix product on the matrix cores -- second half of this file
//   engine 0            packed 10/16-bit fixed-point 2-mer fields read per lane from LDS -- first half
//
// Reference arithmetic being bounded (cscore.c:340-390): for a window without non-ACGT bases
//     s   = fl64( sum_c M[code_c][c] )            (column order)
//     hit = fl64( fl64(s / max_raw) - cutoff ) >= -1e-10
// With x the exact real sum of the same doubles, a reported hit implies
//     x >= T := (cutoff - 1e-10) * max_raw - E,   E = 1e-9 * (1 + sum_c max_b |M[b][c]|)
// (E is ~10^3 times the worst-case fp64 rounding of the W adds, the divide and the subtract).
//
// Engine 0: columns are taken in pairs ("2-mer groups", g = 0..G-1, G = ceil(W/2)); for
// the pair value F_g(code) = M[b0][2g] + M[b1][2g+1] the table stores
//     Q_g(code) = ceil( (max(F_g(code), lo_g) - lo_g) * s )        (an integer >= 0)
// so that  sum_g Q_g >= (x - sum_g lo_g) * s  for every window.  lo_g = max_g - 1.25 * budget
// clamps values so low that the window cannot reach T even with every other group at its
// maximum (budget = best possible sum - T); the clamp only sharpens the resolution.
// A field of FB bits accumulates  B + sum_g Q_g  with  B = 2^(FB-1) - floor((T - sum lo) * s);
// hit  =>  field >= 2^(FB-1)  (its top bit), and the scale s is chosen so the field never
// exceeds 2^FB - 1, hence no carry ever crosses into the neighbouring field of the 32-bit word.
//
// Field width: the pre-filter kernel is bound by LDS bytes, so narrower fields are faster:
//   FB = 10 -> 3 fields per word, a 16-byte table entry serves 6 motifs x {fwd, rev}
//   FB = 16 -> 2 fields per word, 4 motifs per entry
// 10 bits are used whenever the threshold still has >= 2 quantisation levels per group of head
// room (measured on the 579-motif set: 1.12x the true hits instead of 1.01x); otherwise 16.
// Whatever passes is re-scored in fp64 in the reference's order, so the pre-filter decides
// nothing by itself; it only must not lose hits.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "ms_internal.h"

namespace ms {

namespace {

enum QStatus { Q_DEAD = 0, Q_OK = 1, Q_NEEDS_EXACT = 2 };

// e[b][c]: the strand's effective matrix (already flipped for the reverse strand).
QStatus quantize_strand(const double e[4][kMaxFastWidth], int W, double T, int fb, uint16_t q[kMaxGroups][16],
                        double *levels_per_budget) {
    const int G = (W + 1) / 2;
    double F[kMaxGroups][16], maxF[kMaxGroups], minF[kMaxGroups], lo[kMaxGroups];
    double Mx = 0;
    for (int g = 0; g < G; g++) {
        maxF[g] = -INFINITY;
        minF[g] = INFINITY;
        for (int x = 0; x < 16; x++) {
            const int b0 = x & 3, b1 = x >> 2;
            double v = e[b0][2 * g];
            if (2 * g + 1 < W) v += e[b1][2 * g + 1];
            F[g][x] = v;
            maxF[g] = std::max(maxF[g], v);
            minF[g] = std::min(minF[g], v);
        }
        Mx += maxF[g];
    }
    std::memset(q, 0, sizeof(uint16_t) * kMaxGroups * 16);
    const double budget = Mx - T;               // how far below the best window a hit may be
    *levels_per_budget = 1e30;
    if (!(budget >= 0)) return Q_DEAD;          // no N-free window can reach T
    // Clamp a group's deficit (maxF - F) at D > budget: a group that alone overspends the budget
    // is as good as any other such group, but it must still sink the window.  (Clamping AT the
    // budget would let "one hopeless group + all others perfect" sit exactly on the threshold.)
    double scale_abs = 1.0;
    for (int g = 0; g < G; g++) scale_abs += std::fabs(maxF[g]);
    const double D = 1.25 * budget + 1e-7 * scale_abs;
    double sum_lo = 0;
    for (int g = 0; g < G; g++) {
        lo[g] = std::max(minF[g], maxF[g] - D);
        sum_lo += lo[g];
    }
    const double thr_off = T - sum_lo;          // threshold above the clamped floor
    if (!(thr_off > 0)) return Q_NEEDS_EXACT;   // (almost) every window would pass: filter is useless
    const double half = (double) (1u << (fb - 1)), top = (double) ((1u << fb) - 1u);
    double s = half / thr_off;
    if (budget > 0) s = std::min(s, (half - 2.0 - G) / budget);
    for (int attempt = 0; attempt < 200; attempt++, s *= 0.98) {
        uint32_t qq[kMaxGroups][16];
        uint64_t max_sum = 0;
        bool ok = true;
        for (int g = 0; g < G && ok; g++) {
            uint32_t mq = 0;
            for (int x = 0; x < 16; x++) {
                const double v = (F[g][x] - lo[g]) * s;
                double c = v <= 0 ? 0.0 : std::ceil(v * (1 + 1e-12) + 1e-7);
                if (!(c <= top)) { ok = false; break; }
                qq[g][x] = (uint32_t) c;
                mq = std::max(mq, qq[g][x]);
            }
            max_sum += mq;
        }
        if (!ok) continue;
        double fl = std::floor(thr_off * s * (1 - 1e-12) - 1e-7);
        if (fl < 0) fl = 0;
        if (fl > half) continue;
        const uint32_t B = (uint32_t) half - (uint32_t) fl;
        if ((double) B + (double) max_sum > top) continue;
        *levels_per_budget = (D / 1.25) * s;        // resolution of the head room (incl. its absolute floor)
        for (int g = 0; g < G; g++)
            for (int x = 0; x < 16; x++) q[g][x] = (uint16_t) (qq[g][x] + (g == 0 ? B : 0));
        return Q_OK;
    }
    return Q_NEEDS_EXACT;
}

// Can the motif take an integer pre-filter at all, and at which threshold T on the exact real sum
// (header comment: a reported hit implies x >= T)?
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

struct FastMotif {
    int32_t id;
    int32_t G;
    int32_t fb;
    uint16_t fwd[kMaxGroups][16];
    uint16_t rev[kMaxGroups][16];
};

// Quantise both enabled strands at a field width.  Returns false if the motif needs the fp64 path
// at this width (or, with need_levels, if the resolution is too coarse to be a useful filter).
bool quantize_motif(const double *m, int W, double T, int strand_mask, int fb, bool need_levels, FastMotif *fm) {
    std::memset(fm->fwd, 0, sizeof(fm->fwd));
    std::memset(fm->rev, 0, sizeof(fm->rev));
    const int G = (W + 1) / 2;
    for (int strand = 1; strand <= 2; strand <<= 1) {
        if (!(strand_mask & strand)) continue;
        double e[4][kMaxFastWidth];
        for (int b = 0; b < 4; b++)
            for (int c = 0; c < W; c++)
                e[b][c] = strand == 1 ? m[(int64_t) b * W + c] : m[(int64_t) (3 - b) * W + (W - 1 - c)];   // cscore.c:351
        double levels = 0;
        const QStatus st = quantize_strand(e, W, T, fb, strand == 1 ? fm->fwd : fm->rev, &levels);
        if (st == Q_NEEDS_EXACT) return false;
        if (need_levels && st == Q_OK && levels < 2.0 * G) return false;
    }
    fm->G = G;
    fm->fb = fb;
    return true;
}

}  // namespace

int build_plan(const double *values, const int64_t *val_off, const int32_t *widths,
               const double *cutoffs, const double *max_raw, int32_t n_pwms, int strand_mask,
               size_t lds_budget, int min_field_bits, PrefilterPlan *plan) {
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
        if (ok) {
            ok = (min_field_bits <= 10 && quantize_motif(m, W, T, strand_mask, 10, true, &fm)) ||
                 quantize_motif(m, W, T, strand_mask, 16, false, &fm);
        }
        if (ok) fast.push_back(fm);
        else plan->exact_motifs.push_back(p);
    }

    // groups of 2 * (32 / FB) motifs sharing a 16-byte table entry; same field width together,
    // narrow to wide so a group's 2-mer count wastes little
    std::stable_sort(fast.begin(), fast.end(), [](const FastMotif &a, const FastMotif &b) {
        return a.fb != b.fb ? a.fb < b.fb : a.G < b.G;
    });
    std::vector<size_t> group_off16(1, 0);
    size_t i = 0;
    while (i < fast.size()) {
        const int fb = fast[i].fb;
        const int nf = 32 / fb, nm = 2 * nf;              // fields per word, motifs per group
        const int32_t grp = (int32_t) plan->group_G.size();
        int G = 0, cnt = 0;
        for (; cnt < nm && i + cnt < fast.size() && fast[i + cnt].fb == fb; cnt++) G = std::max(G, (int) fast[i + cnt].G);
        plan->group_G.push_back(G);
        plan->group_fb.push_back(fb);
        plan->group_motifs.resize((size_t) (grp + 1) * kGroupSlots, -1);
        const size_t off16 = group_off16.back();
        plan->tables.resize((off16 + (size_t) G * 16) * 4, 0u);
        for (int j = 0; j < cnt; j++) {
            const FastMotif &fm = fast[i + j];
            plan->group_motifs[(size_t) grp * kGroupSlots + j] = fm.id;
            plan->fast_motifs.push_back(fm.id);
            // motif j: forward field n = 2j, reverse n = 2j+1; field n lives in word n & 3 at bit (n >> 2) * FB
            for (int sd = 0; sd < 2; sd++) {
                const int n = 2 * j + sd, word = n & 3, shift = (n >> 2) * fb;
                for (int g = 0; g < fm.G; g++)               // groups beyond the motif's own G stay 0
                    for (int x = 0; x < 16; x++)
                        plan->tables[(off16 + (size_t) g * 16 + x) * 4 + word] |=
                            (uint32_t) (sd == 0 ? fm.fwd[g][x] : fm.rev[g][x]) << shift;
            }
        }
        group_off16.push_back(off16 + (size_t) G * 16);
        plan->lds_bytes_per_position += (int64_t) G * 16;
        i += cnt;
    }
    const int32_t n_groups = (int32_t) plan->group_G.size();

    // LDS tiles of equal work (work ~ table bytes)
    if (n_groups > 0) {
        const size_t total16 = group_off16[n_groups];
        const size_t budget16 = std::max<size_t>(lds_budget / 16, (size_t) kMaxGroups * 16);
        const size_t n_tiles = (total16 + budget16 - 1) / budget16;
        const size_t target16 = (total16 + n_tiles - 1) / n_tiles;
        int32_t q = 0;
        while (q < n_groups) {
            TileDesc t;
            std::memset(&t, 0, sizeof(t));
            t.table_off16 = (uint32_t) group_off16[q];
            t.first_group = q;
            size_t used = 0;
            while (q < n_groups) {
                const size_t need = (size_t) plan->group_G[q] * 16;
                if (used > 0 && (used + need > budget16 || used >= target16)) break;
                if (t.n_classes == 0 || t.cls[t.n_classes - 1].G != plan->group_G[q] ||
                    t.cls[t.n_classes - 1].fb != plan->group_fb[q]) {
                    t.cls[t.n_classes].G = plan->group_G[q];
                    t.cls[t.n_classes].fb = plan->group_fb[q];
                    t.cls[t.n_classes].n_groups = 0;
                    t.cls[t.n_classes].base16 = (uint32_t) used;
                    t.cls[t.n_classes].first_group = q;
                    t.n_classes++;
                }
                t.cls[t.n_classes - 1].n_groups++;
                used += need;
                q++;
            }
            t.table_len16 = (uint32_t) used;
            plan->tiles.push_back(t);
        }
    }
    return MS_OK;
}

// ------------------------------------------------------------------ engine 1: int8 / MFMA --
//
// Per strand, with e[b][c] the effective matrix, hi_c = max_b e[b][c] and the deficit
// d_c(b) = hi_c - e[b][c] >= 0:   x = sum hi_c - sum_c d_c(code_c), so
//     hit  =>  sum_c d_c(code_c) <= budget := sum hi_c - T.
// Deficits are quantised DOWN, dq_c(b) = min(floor(d_c(b) * s), Bq + 1) with Bq = floor(budget * s)
// (both with a hair of slack against the fp64 rounding of the products), so a hit implies
// sum dq <= Bq; a clamped column alone already exceeds Bq, so clamping changes no decision that
// matters.  The table stores v_c(b) = t_c - dq_c(b) with sum_c t_c = Bq:
//     acc = sum_c v_c(code_c) = Bq - sum dq >= 0   <=>   candidate  (sign bit of the i32 result).
// int8 range: v_c in [t_c - cap_c, t_c] with cap_c the largest dq of the column, so any
// t_c in [cap_c - 128, 127] works; the largest Bq <= 254 for which such t_c can sum to Bq is taken
// (>= 127 always: one column carries Bq, the others 0).  Columns past the motif's width hold 0.
namespace {

struct I8Strand {
    int8_t v[kMaxFastWidth][4];
    int levels;                     // Bq: quantisation levels of the budget (0: strand can never hit)
};

// Returns false if the strand needs the fp64 path (threshold so low that filtering is pointless).
bool quantize_strand_i8(const double e[4][kMaxFastWidth], int W, double T, I8Strand *out) {
    std::memset(out->v, 0, sizeof(out->v));
    out->levels = 0;
    double hi[kMaxFastWidth], Mx = 0, lowest = 0;
    for (int c = 0; c < W; c++) {
        hi[c] = std::max(std::max(e[0][c], e[1][c]), std::max(e[2][c], e[3][c]));
        Mx += hi[c];
        lowest += std::min(std::min(e[0][c], e[1][c]), std::min(e[2][c], e[3][c]));
    }
    const double budget = Mx - T;
    if (!(budget >= 0)) {                       // dead: no N-free window reaches T -> acc = -1 everywhere
        for (int b = 0; b < 4; b++) out->v[0][b] = -1;
        return true;
    }
    if (!(T > lowest)) return false;            // every window passes
    // Levels of the budget.  The matrix-core kernel is power-limited: operand bytes of small magnitude let the
    // chip hold a higher clock, and with <= 127 levels the best base of every column is 0x00 and a clamped one 0x80.  Motifs of
    // up to 16 columns lose little resolution at 127 levels (the budget is shared by few columns); wider ones keep 254.
    // MS_MEASURE=1 MS_PF_BQ_MAX=n forces one value for all motifs (A/B: profiles/r02_operand_activity_bq127.log).
    int bq_max = W <= 16 ? 127 : 254;
    if (const char *e = measure_env("MS_PF_BQ_MAX")) bq_max = std::max(1, std::min(254, atoi(e)));
    for (int Bq = bq_max; Bq >= 1; Bq--) {
        const double s = budget > 0 ? ((double) Bq + 0.5) / budget : 1e300;
        int dq[kMaxFastWidth][4], cap[kMaxFastWidth];
        long lo_sum = 0;
        for (int c = 0; c < W; c++) {
            cap[c] = 0;
            for (int b = 0; b < 4; b++) {
                const double d = hi[c] - e[b][c];
                double q = d <= 0 ? 0.0 : std::floor(std::min(d * s * (1 - 1e-12) - 1e-7, 1e6));
                if (!(q >= 0)) q = 0;
                dq[c][b] = (int) std::min<double>(q, Bq + 1);
                cap[c] = std::max(cap[c], dq[c][b]);
            }
            lo_sum += cap[c] - 128;
        }
        if (lo_sum > Bq) continue;              // the offsets t_c cannot sum to Bq inside int8
        long rest = Bq - lo_sum;                // >= 0; hand it out over the columns
        for (int c = 0; c < W; c++) {
            const int lo_t = cap[c] - 128;
            const int add = (int) std::min<long>(rest, 127 - lo_t);
            const int t = lo_t + add;
            rest -= add;
            for (int b = 0; b < 4; b++) out->v[c][b] = (int8_t) (t - dq[c][b]);
        }
        if (rest != 0) continue;                // (cannot happen: sum of 127 >= Bq)
        out->levels = Bq;
        return true;
    }
    return false;
}

// Engine 2 (ms_internal.h): the same deficits, stored as Walsh coefficients.  Per column the integers
// w(b) = -dq(b) <= 0 are nudged UP (never down: the bound only loosens; never above 0: no window gains a bonus)
// by the smallest amounts that make c = H w / 4 integral -- w(1), w(2), w(3) of one parity and the sum
// divisible by 4 -- and then
//     w'(b) = c0 + c1*s1(b) + c2*s2(b) + c3*s1(b)*s2(b)      exactly,
// with c1..c3 in int8 (|c| <= 127 at dq <= 254) and the c0 summed into the row bias next to Bq:
//     acc = bias + sum_c (c1 s1 + c2 s2 + c3 s1 s2)(base_c) = Bq + sum_c w'_c(base_c) >= Bq - sum dq.
struct W2Strand {
    int8_t c[kMaxFastWidth][3];
    int32_t bias;
    int levels;
};

bool quantize_strand_w2(const double e[4][kMaxFastWidth], int W, double T, W2Strand *out) {
    std::memset(out->c, 0, sizeof(out->c));
    out->bias = -1;                             // dead until proven otherwise: acc = -1 everywhere
    out->levels = 0;
    double hi[kMaxFastWidth], Mx = 0, lowest = 0;
    for (int c = 0; c < W; c++) {
        hi[c] = std::max(std::max(e[0][c], e[1][c]), std::max(e[2][c], e[3][c]));
        Mx += hi[c];
        lowest += std::min(std::min(e[0][c], e[1][c]), std::min(e[2][c], e[3][c]));
    }
    const double budget = Mx - T;
    if (!(budget >= 0)) return true;            // no N-free window reaches T
    if (!(T > lowest)) return false;            // every window passes
    for (int Bq = 251; Bq >= 1; Bq--) {
        const double s = budget > 0 ? ((double) Bq + 0.5) / budget : 1e300;
        int8_t cc[kMaxFastWidth][3];
        long bias = Bq;
        bool ok = true;
        for (int c = 0; c < W && ok; c++) {
            // a deficit beyond the budget is stored as Bq + 4: it may be nudged by up to 3 and still sinks the window alone
            int w[4];
            bool clamped[4];
            for (int b = 0; b < 4; b++) {
                const double d = hi[c] - e[b][c];
                double q = d <= 0 ? 0.0 : std::floor(std::min(d * s * (1 - 1e-12) - 1e-7, 1e6));
                if (!(q >= 0)) q = 0;
                clamped[b] = q > Bq;
                w[b] = clamped[b] ? -(Bq + 4) : -(int) q;
            }
            // smallest upward nudges (none above 0, so no window ever gains a bonus; clamped entries are free):
            // w1, w2, w3 of one parity, sum = 0 mod 4
            int best = 1 << 30, bd[4] = {0, 0, 0, 0};
            for (int code = 0; code < 256; code++) {
                const int dd[4] = {code & 3, (code >> 2) & 3, (code >> 4) & 3, (code >> 6) & 3};
                const int v0 = w[0] + dd[0], v1 = w[1] + dd[1], v2 = w[2] + dd[2], v3 = w[3] + dd[3];
                if (v0 > 0 || v1 > 0 || v2 > 0 || v3 > 0) continue;
                if (((v1 ^ v2) & 1) || ((v1 ^ v3) & 1)) continue;
                if ((v0 + v1 + v2 + v3) & 3) continue;
                int cost = 0;
                for (int b = 0; b < 4; b++) cost += clamped[b] ? 0 : dd[b];
                if (cost < best) { best = cost; for (int b = 0; b < 4; b++) bd[b] = dd[b]; }
            }
            if (best == (1 << 30)) { ok = false; break; }              // (values too close to 0 to fix the parities: try a coarser scale)
            const int v0 = w[0] + bd[0], v1 = w[1] + bd[1], v2 = w[2] + bd[2], v3 = w[3] + bd[3];
            // base code b: s1 = +1 for b in {0, 2}, -1 for {1, 3};  s2 = +1 for {0, 1}, -1 for {2, 3}
            const int c0 = (v0 + v1 + v2 + v3) / 4, c1 = (v0 - v1 + v2 - v3) / 4, c2 = (v0 + v1 - v2 - v3) / 4, c3 = (v0 - v1 - v2 + v3) / 4;
            if (c1 < -127 || c1 > 127 || c2 < -127 || c2 > 127 || c3 < -127 || c3 > 127) { ok = false; break; }
            cc[c][0] = (int8_t) c1; cc[c][1] = (int8_t) c2; cc[c][2] = (int8_t) c3;
            bias += c0;
        }
        if (!ok) continue;
        if (bias < -64 * 127 - 32 || bias > 64 * 127 + 32) continue;       // must fit 64 * a_hi + a_lo with int8 parts
        for (int c = 0; c < W; c++) for (int t = 0; t < 3; t++) out->c[c][t] = cc[c][t];
        out->bias = (int32_t) bias;
        out->levels = Bq;
        return true;
    }
    return false;
}

// Engine 3 (ms_internal.h): the same deficits on the fp6 e2m3 grid, in units of 1/8.
struct F6Strand {
    int8_t u[kMaxFastWidth][4];     // t_c - dq_c(b)
    int levels;
};

inline int f6_grid_floor(int q) { return q <= 16 ? q : (q <= 32 ? (q & ~1) : (q & ~3)); }

bool quantize_strand_f6(const double e[4][kMaxFastWidth], int W, double T, F6Strand *out) {
    std::memset(out->u, 0, sizeof(out->u));
    out->levels = 0;
    double hi[kMaxFastWidth], Mx = 0, lowest = 0;
    for (int c = 0; c < W; c++) {
        hi[c] = std::max(std::max(e[0][c], e[1][c]), std::max(e[2][c], e[3][c]));
        Mx += hi[c];
        lowest += std::min(std::min(e[0][c], e[1][c]), std::min(e[2][c], e[3][c]));
    }
    const double budget = Mx - T;
    if (!(budget >= 0)) {                       // dead: no N-free window reaches T -> acc = -1/8 everywhere
        for (int b = 0; b < 4; b++) out->u[0][b] = -1;
        return true;
    }
    if (!(T > lowest)) return false;            // every window passes
    if (W < 4) return false;                    // the offsets need four columns
    const int Bq = kF6Levels;
    const double s = budget > 0 ? ((double) Bq + 0.5) / budget : 1e300;
    static const int t[4] = {16, 16, 16, 8};
    for (int c = 0; c < W; c++)
        for (int b = 0; b < 4; b++) {
            const double d = hi[c] - e[b][c];
            double q = d <= 0 ? 0.0 : std::floor(std::min(d * s * (1 - 1e-12) - 1e-7, 1e6));
            if (!(q >= 0)) q = 0;
            const int dq = q > Bq ? 60 : f6_grid_floor((int) q);          // down to the grid; beyond the budget: 60 sinks the window alone
            const int u = (c < 4 ? t[c] : 0) - dq;
            if (!f6_representable(u)) return false;                       // (cannot happen: offsets are multiples of 4)
            out->u[c][b] = (int8_t) u;
        }
    out->levels = Bq;
    return true;
}

struct FastMotifI8 {
    int32_t id;
    int32_t W;
    I8Strand strand[2];
    W2Strand w2[2];
    F6Strand f6[2];
};

}  // namespace

int build_plan_mfma(const double *values, const int64_t *val_off, const int32_t *widths,
                    const double *cutoffs, const double *max_raw, int32_t n_pwms, int strand_mask,
                    size_t lds_budget, int engine, PrefilterPlan *plan) {
    *plan = PrefilterPlan();
    plan->strand_mask = strand_mask;
    plan->engine = engine == 2 ? 2 : engine == 3 ? 3 : 1;
    const int cols = plan->engine == 2 ? kW2Cols : plan->engine == 3 ? kF6Cols : 8;              // motif columns per k-block
    const int max_w = plan->engine == 2 ? kW2MaxWidth : kMaxFastWidth;
    const size_t kb_bytes = plan->engine == 3 ? (size_t) kF6BytesPerKb : (size_t) kMfmaRowTileBytesPerKb;
    std::vector<FastMotifI8> fast;
    fast.reserve(n_pwms);
    for (int32_t p = 0; p < n_pwms; p++) {
        const int W = widths[p];
        const double *m = values + val_off[p];
        double T = 0;
        bool ok = W <= max_w && filter_threshold(m, W, cutoffs[p], max_raw[p], &T);
        FastMotifI8 fm;
        fm.id = p;
        fm.W = W;
        for (int sd = 0; ok && sd < 2; sd++) {
            std::memset(&fm.strand[sd], 0, sizeof(I8Strand));
            std::memset(&fm.w2[sd], 0, sizeof(W2Strand));
            std::memset(&fm.f6[sd], 0, sizeof(F6Strand));
            for (int b = 0; b < 4; b++) fm.strand[sd].v[0][b] = -1;      // never a candidate unless quantised below
            for (int b = 0; b < 4; b++) fm.f6[sd].u[0][b] = -1;
            fm.w2[sd].bias = -1;
            if (!(strand_mask & (1 << sd))) continue;                    // strand not asked for
            double e[4][kMaxFastWidth];
            for (int b = 0; b < 4; b++)
                for (int c = 0; c < W; c++)
                    e[b][c] = sd == 0 ? m[(int64_t) b * W + c] : m[(int64_t) (3 - b) * W + (W - 1 - c)];   // cscore.c:351
            ok = plan->engine == 2 ? quantize_strand_w2(e, W, T, &fm.w2[sd])
                 : plan->engine == 3 ? quantize_strand_f6(e, W, T, &fm.f6[sd]) : quantize_strand_i8(e, W, T, &fm.strand[sd]);
        }
        if (ok) fast.push_back(fm);
        else plan->exact_motifs.push_back(p);
    }
    std::stable_sort(fast.begin(), fast.end(), [cols](const FastMotifI8 &a, const FastMotifI8 &b) {
        const int ka = (a.W + cols - 1) / cols, kb = (b.W + cols - 1) / cols;
        return ka != kb ? ka < kb : a.W < b.W;
    });

    // row tiles of 16 motifs (2 table groups), narrow to wide
    const size_t n_rt = (fast.size() + 15) / 16;
    std::vector<int> rt_kb(n_rt, 0);
    std::vector<size_t> rt_off(n_rt + 1, 0);
    for (size_t t = 0; t < n_rt; t++) {
        for (size_t j = 16 * t; j < std::min(fast.size(), 16 * (t + 1)); j++) rt_kb[t] = std::max(rt_kb[t], (fast[j].W + cols - 1) / cols);
        rt_off[t + 1] = rt_off[t] + (size_t) rt_kb[t] * kb_bytes;
    }
    std::vector<uint8_t> bytes(rt_off[n_rt], 0);
    plan->group_motifs.assign(2 * n_rt * kGroupSlots, -1);
    plan->group_G.assign(2 * n_rt, 0);
    plan->group_fb.assign(2 * n_rt, 8);
    for (size_t t = 0; t < n_rt; t++) {
        uint8_t *tab = bytes.data() + rt_off[t];
        for (int h = 0; h < 2; h++) {
            const size_t grp = 2 * t + h;
            plan->group_G[grp] = rt_kb[t];
            for (int slot = 0; slot < kGroupSlots; slot++) {
                const size_t j = 16 * t + 8 * h + slot;
                for (int sd = 0; sd < 2; sd++) {
                    const int row = mfma_row_of(h, 2 * slot + sd);
                    if (plan->engine == 3) {
                        // empty slot / dead strand: -1/8 at column 0 for every base -> never a candidate; columns past W stay +0
                        for (int c = 0; c < kF6Cols * rt_kb[t] && c < kMaxFastWidth; c++)
                            for (int b = 0; b < 4; b++) {
                                int u = 0;
                                if (j >= fast.size()) u = c == 0 ? -1 : 0;
                                else if (c < fast[j].W) u = fast[j].f6[sd].u[c][b];
                                f6_put(tab, c / kF6Cols, row, c % kF6Cols, b, f6_code(u));
                            }
                        continue;
                    }
                    if (plan->engine == 2) {
                        // bias = 64 * a_hi + a_lo in the spare bytes of k-block 0; an empty slot or dead strand: bias -1, no coefficients
                        const int32_t bias = j < fast.size() ? fast[j].w2[sd].bias : -1;
                        int a_hi = (int) std::lround((double) bias / 64.0);
                        a_hi = std::max(-127, std::min(127, a_hi));
                        const int a_lo = bias - 64 * a_hi;               // |a_lo| <= 32 by the quantiser's range check
                        tab[mfma2_spare_index(row, 0)] = (uint8_t) (int8_t) a_hi;
                        tab[mfma2_spare_index(row, 1)] = (uint8_t) (int8_t) a_lo;
                        if (j < fast.size())
                            for (int c = 0; c < fast[j].W; c++)
                                for (int s3 = 0; s3 < 3; s3++)
                                    tab[mfma2_byte_index(c / cols, row, c % cols, s3)] = (uint8_t) fast[j].w2[sd].c[c][s3];
                        continue;
                    }
                    if (j >= fast.size()) {                      // empty slot: never a candidate
                        for (int b = 0; b < 4; b++) tab[mfma_byte_index(0, row, 0, b)] = (uint8_t) (int8_t) -1;
                        continue;
                    }
                    const FastMotifI8 &fm = fast[j];
                    // an all-zero row would flag every window (acc = 0): a quantised strand never is one
                    // (dead strands carry -1), and columns past W stay 0
                    for (int c = 0; c < 8 * rt_kb[t] && c < kMaxFastWidth; c++)
                        for (int b = 0; b < 4; b++)
                            tab[mfma_byte_index(c >> 3, row, c & 7, b)] = (uint8_t) (c < fm.W ? fm.strand[sd].v[c][b] : 0);
                }
                if (j < fast.size()) {
                    plan->group_motifs[grp * kGroupSlots + slot] = fast[j].id;
                }
            }
        }
        plan->lds_bytes_per_position += (int64_t) rt_kb[t] * (int64_t) kb_bytes / 64;     // A-operand bytes per window start (2 x 32 windows share a read)
    }
    for (const FastMotifI8 &fm : fast) plan->fast_motifs.push_back(fm.id);
    plan->tables.resize(bytes.size() / 4);
    if (!bytes.empty()) std::memcpy(plan->tables.data(), bytes.data(), bytes.size());

    // LDS tiles (whole row tiles; work ~ bytes), classes = runs of equal k-block count
    if (n_rt > 0) {
        const size_t total = rt_off[n_rt];
        const size_t budget = std::max<size_t>(lds_budget, (size_t) 4 * kb_bytes);
        const size_t n_tiles = (total + budget - 1) / budget;
        const size_t target = (total + n_tiles - 1) / n_tiles;
        size_t q = 0;
        while (q < n_rt) {
            TileDesc t;
            std::memset(&t, 0, sizeof(t));
            t.table_off16 = (uint32_t) (rt_off[q] / 16);
            t.first_group = (int32_t) (2 * q);
            size_t used = 0;
            while (q < n_rt) {
                const size_t need = (size_t) rt_kb[q] * kb_bytes;
                if (used > 0 && (used + need > budget || used >= target)) break;
                if (t.n_classes == 0 || t.cls[t.n_classes - 1].G != rt_kb[q]) {
                    ClassDesc &cd = t.cls[t.n_classes++];
                    cd.G = rt_kb[q];
                    cd.fb = 8;
                    cd.n_groups = 0;
                    cd.base16 = (uint32_t) (used / 16);
                    cd.first_group = (int32_t) (2 * q);
                }
                t.cls[t.n_classes - 1].n_groups++;
                used += need;
                q++;
            }
            t.table_len16 = (uint32_t) (used / 16);
            plan->tiles.push_back(t);
        }
    }
    return MS_OK;
}

}  // namespace ms
