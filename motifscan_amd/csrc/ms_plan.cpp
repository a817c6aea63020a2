// ms_plan.cpp -- host side of the integer pre-filter: quantise every PWM into 16-bit 2-mer
// tables that can NEVER miss a window the reference would report, and cut the tables into
// LDS tiles.
//
// Reference arithmetic being bounded (cscore.c:340-390): for a window without non-ACGT bases
//     s   = fl64( sum_c M[code_c][c] )            (column order)
//     hit = fl64( fl64(s / max_raw) - cutoff ) >= -1e-10
// With x the exact real sum of the same doubles, a reported hit implies
//     x >= T := (cutoff - 1e-10) * max_raw - E,   E = 1e-9 * (1 + sum_c max_b |M[b][c]|)
// (E is ~10^3 times the worst-case fp64 rounding of the W adds, the divide and the subtract).
//
// Pre-filter: columns are taken in pairs ("2-mer groups", g = 0..G-1, G = ceil(W/2)); for
// the pair value F_g(code) = M[b0][2g] + M[b1][2g+1] the table stores
//     Q_g(code) = ceil( (max(F_g(code), lo_g) - lo_g) * s )        (an integer >= 0)
// so that  sum_g Q_g >= (x - sum_g lo_g) * s  for every window.  lo_g = max_g - 1.25 * budget
// clamps values so low that the window cannot reach T even with every other group at its
// maximum (budget = best possible sum - T); the clamp only sharpens the 16-bit resolution.
// A 16-bit field accumulates  B + sum_g Q_g  with  B = 0x8000 - floor((T - sum lo) * s);
// hit  =>  field >= 0x8000, and the scale s is chosen so the field never exceeds 0xFFFF.
// Forward and reverse fields of one motif share a 32-bit word (no carry can cross).
// Whatever passes is re-scored in fp64 in the reference's order, so the pre-filter decides
// nothing by itself; it only must not lose hits.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "ms_internal.h"

namespace ms {

namespace {

enum QStatus { Q_DEAD = 0, Q_OK = 1, Q_NEEDS_EXACT = 2 };

// e[b][c]: the strand's effective matrix (already flipped for the reverse strand).
QStatus quantize_strand(const double e[4][kMaxFastWidth], int W, double T, uint16_t q[kMaxGroups][16]) {
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
    double s = 32768.0 / thr_off;
    if (budget > 0) s = std::min(s, (32766.0 - G) / budget);
    for (int attempt = 0; attempt < 200; attempt++, s *= 0.98) {
        uint32_t qq[kMaxGroups][16];
        uint64_t max_sum = 0;
        bool ok = true;
        for (int g = 0; g < G && ok; g++) {
            uint32_t mq = 0;
            for (int x = 0; x < 16; x++) {
                const double v = (F[g][x] - lo[g]) * s;
                double c = v <= 0 ? 0.0 : std::ceil(v * (1 + 1e-12) + 1e-7);
                if (!(c <= 65535.0)) { ok = false; break; }
                qq[g][x] = (uint32_t) c;
                mq = std::max(mq, qq[g][x]);
            }
            max_sum += mq;
        }
        if (!ok) continue;
        double fl = std::floor(thr_off * s * (1 - 1e-12) - 1e-7);
        if (fl < 0) fl = 0;
        if (fl > 32768.0) continue;
        const uint32_t B = 32768u - (uint32_t) fl;
        if ((uint64_t) B + max_sum > 65535u) continue;
        for (int g = 0; g < G; g++)
            for (int x = 0; x < 16; x++) q[g][x] = (uint16_t) (qq[g][x] + (g == 0 ? B : 0));
        return Q_OK;
    }
    return Q_NEEDS_EXACT;
}

struct FastMotif {
    int32_t id;
    int32_t G;
    uint32_t words[kMaxGroups][16];   // lo16 = forward field, hi16 = reverse field
};

}  // namespace

int build_plan(const double *values, const int64_t *val_off, const int32_t *widths,
               const double *cutoffs, const double *max_raw, int32_t n_pwms, int strand_mask,
               size_t lds_budget, PrefilterPlan *plan) {
    *plan = PrefilterPlan();
    plan->strand_mask = strand_mask;
    std::vector<FastMotif> fast;
    fast.reserve(n_pwms);
    for (int32_t p = 0; p < n_pwms; p++) {
        const int W = widths[p];
        const double *m = values + val_off[p];
        bool ok = W >= 1 && W <= kMaxFastWidth && std::isfinite(max_raw[p]) && max_raw[p] > 0 &&
                  std::isfinite(cutoffs[p]);
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
        FastMotif fm;
        if (ok) {
            const double E = 1e-9 * (1.0 + abs_sum);
            const double T = (cutoffs[p] - 1e-10) * max_raw[p] - E;
            fm.id = p;
            fm.G = (W + 1) / 2;
            std::memset(fm.words, 0, sizeof(fm.words));
            for (int strand = 1; strand <= 2 && ok; strand <<= 1) {
                if (!(strand_mask & strand)) continue;
                double e[4][kMaxFastWidth];
                for (int b = 0; b < 4; b++)
                    for (int c = 0; c < W; c++)
                        e[b][c] = strand == 1 ? m[(int64_t) b * W + c]
                                              : m[(int64_t) (3 - b) * W + (W - 1 - c)];   // cscore.c:351
                uint16_t q[kMaxGroups][16];
                const QStatus st = quantize_strand(e, W, T, q);
                if (st == Q_NEEDS_EXACT) { ok = false; break; }
                for (int g = 0; g < fm.G; g++)
                    for (int x = 0; x < 16; x++)
                        fm.words[g][x] |= (uint32_t) q[g][x] << (strand == 1 ? 0 : 16);
            }
        }
        if (ok) fast.push_back(fm);
        else plan->exact_motifs.push_back(p);
    }

    // quads of four motifs, narrow to wide so a quad's group count wastes little
    std::stable_sort(fast.begin(), fast.end(), [](const FastMotif &a, const FastMotif &b) { return a.G < b.G; });
    const int32_t n_quads = (int32_t) ((fast.size() + 3) / 4);
    plan->quad_motifs.assign((size_t) n_quads * 4, -1);
    plan->quad_G.assign(n_quads, 0);
    std::vector<size_t> quad_off16(n_quads + 1, 0);
    for (int32_t q = 0; q < n_quads; q++) {
        int G = 0;
        for (int k = 0; k < 4; k++) {
            const size_t i = (size_t) q * 4 + k;
            if (i < fast.size()) {
                plan->quad_motifs[i] = fast[i].id;
                plan->fast_motifs.push_back(fast[i].id);
                G = std::max(G, (int) fast[i].G);
            }
        }
        plan->quad_G[q] = G;
        quad_off16[q + 1] = quad_off16[q] + (size_t) G * 16;
        plan->lds_bytes_per_position += (int64_t) G * 16;
    }
    plan->tables.assign(quad_off16[n_quads] * 4, 0u);
    for (int32_t q = 0; q < n_quads; q++)
        for (int k = 0; k < 4; k++) {
            const size_t i = (size_t) q * 4 + k;
            if (i >= fast.size()) continue;
            for (int g = 0; g < fast[i].G; g++)            // groups beyond the motif's own G stay 0
                for (int x = 0; x < 16; x++)
                    plan->tables[(quad_off16[q] + (size_t) g * 16 + x) * 4 + k] = fast[i].words[g][x];
        }

    // LDS tiles of equal work (work ~ table bytes)
    if (n_quads > 0) {
        const size_t total16 = quad_off16[n_quads];
        const size_t budget16 = std::max<size_t>(lds_budget / 16, (size_t) kMaxGroups * 16);
        const size_t n_tiles = (total16 + budget16 - 1) / budget16;
        const size_t target16 = (total16 + n_tiles - 1) / n_tiles;
        int32_t q = 0;
        while (q < n_quads) {
            TileDesc t;
            std::memset(&t, 0, sizeof(t));
            t.table_off16 = (uint32_t) quad_off16[q];
            t.first_quad = q;
            size_t used = 0;
            while (q < n_quads) {
                const size_t need = (size_t) plan->quad_G[q] * 16;
                if (used > 0 && (used + need > budget16 || used >= target16)) break;
                if (t.n_classes == 0 || t.cls[t.n_classes - 1].G != plan->quad_G[q]) {
                    t.cls[t.n_classes].G = plan->quad_G[q];
                    t.cls[t.n_classes].n_quads = 0;
                    t.n_classes++;
                }
                t.cls[t.n_classes - 1].n_quads++;
                used += need;
                q++;
            }
            t.table_len16 = (uint32_t) used;
            plan->tiles.push_back(t);
        }
    }
    return MS_OK;
}

}  // namespace ms
