// ms_device.h -- device-side helpers shared by the kernel translation units (ms_kernels.hip, ms_tail.hip): reading
// the packed sequence, position -> region, and the fp64 window scorer in the reference's order of operations.
#pragma once
#include "ms_kernels.h"

namespace ms {

// (the fp64 stage is bound by the texture addresser's instruction rate -- profiles/r03t_rescore_ta.log -- so a window's three code
// words / two mask words come with ONE load each: word-aligned 12- and 8-byte loads)
struct __attribute__((packed, aligned(4))) Words3 { uint32_t a, b, c; };
struct __attribute__((packed, aligned(4))) Words2 { uint32_t a, b; };

__device__ __forceinline__ uint64_t code_window(const uint32_t *__restrict__ codes, int64_t g) {
    const int64_t wi = g >> 4;
    const uint32_t sh = ((uint32_t) g & 15u) * 2u;
    const Words3 w3 = *reinterpret_cast<const Words3 *>(codes + wi);
    const uint32_t w0 = w3.a, w1 = w3.b, w2 = w3.c;
    const uint64_t lo = ((uint64_t) w1 << 32) | w0;
    return sh ? (lo >> sh) | ((uint64_t) w2 << (64u - sh)) : lo;
}

__device__ __forceinline__ uint32_t n_window(const uint32_t *__restrict__ nmask, int64_t g) {
    const int64_t wi = g >> 5;
    const uint32_t sh = (uint32_t) g & 31u;
    const Words2 w2 = *reinterpret_cast<const Words2 *>(nmask + wi);
    const uint32_t w0 = w2.a, w1 = w2.b;
    return sh ? (w0 >> sh) | (w1 << (32u - sh)) : w0;
}

__device__ __forceinline__ uint32_t low_mask(int w) { return w >= 32 ? 0xFFFFFFFFu : ((1u << w) - 1u); }

// region r with offsets[r] <= g < offsets[r+1]  (empty regions are skipped by construction).
// blk2reg[g >> 6] is the region of position (g & ~63): a short forward walk finds g's region for
// ordinary region lengths; tiny regions fall back to a binary search from there.
__device__ __forceinline__ int64_t find_region(const DevSeq &S, int64_t g) {
    int64_t lo = S.blk2reg[g >> 6];
#pragma unroll 1
    for (int step = 0; step < 4; step++) {
        if (S.offsets[lo + 1] > g) return lo;
        lo++;
    }
    int64_t hi = S.R;                 // invariant: offsets[lo] <= g, answer in [lo, hi)
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (S.offsets[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ int64_t find_region_bsearch(const int64_t *__restrict__ offsets, int64_t R, int64_t g) {
    int64_t lo = 0, hi = R;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

// fp64 scores of one window in the reference's order: c = 0..W-1, forward adds M[row][c],
// reverse adds M[3-row][W-1-c], non-ACGT adds nothing (cscore.c:345-354).  The table entries of
// eight columns are fetched together (independent loads), then added strictly in column order.
__device__ __forceinline__ void score_window(const DevSeq &S, const double2 *__restrict__ tab, int W,
                                             int64_t g, double &fwd, double &rev) {
    fwd = 0.0;
    rev = 0.0;
    for (int c0 = 0; c0 < W; c0 += 32) {
        const uint64_t cw = code_window(S.codes, g + c0);
        const uint32_t nw = n_window(S.nmask, g + c0);
        const int n = (W - c0) < 32 ? (W - c0) : 32;
        for (int c1 = 0; c1 < n; c1 += 8) {
            double2 t[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int c = c1 + k;
                const uint32_t b = (uint32_t) (cw >> (2 * (c & 31))) & 3u;
                t[k] = tab[(c0 + (c < n ? c : n - 1)) * 4 + b];               // clamped: always a valid entry
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int c = c1 + k;
                if (c < n && !((nw >> c) & 1u)) {
                    fwd += t[k].x;
                    rev += t[k].y;
                }
            }
        }
    }
}

// The same for W <= 32 with the lane's code / mask windows already in registers, without a branch: tab_bytes = byte offset of the
// motif's table in tab2 (every offset fits 32 bits: DevPwm::tab32), and a column that adds nothing -- a non-ACGT base, or a column
// past the motif in the last batch of eight -- reads tab2's closing all-zero entry instead and ADDS it: x + (+0.0) = x for every x
// a running sum that started at +0.0 can hold (it is never -0.0: that takes two -0.0 operands).  Per column: two bit-field extracts,
// a shift-add, a bit-field insert, the load (uniform base + 32-bit offset) and the two fp64 adds, in column order.
template <int N>
__device__ __forceinline__ void score_columns(const char *__restrict__ base, uint32_t t0, uint32_t zero_bytes, uint32_t bits, uint32_t sk, double &fwd, double &rev) {
    double2 t[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        const uint32_t off = t0 + (uint32_t) k * 64u + ((bits >> (2 * k)) & 3u) * 16u;
        const uint32_t sel = (uint32_t) -(int32_t) ((sk >> k) & 1u);          // all ones: the zero entry
        t[k] = *reinterpret_cast<const double2 *>(base + ((off & ~sel) | (zero_bytes & sel)));
    }
#pragma unroll
    for (int k = 0; k < N; k++) {
        fwd += t[k].x;
        rev += t[k].y;
    }
}
__device__ __forceinline__ void score_window32(const double2 *__restrict__ tab2, uint32_t tab_bytes, uint32_t zero_bytes, int W, uint64_t cw, uint32_t nw,
                                               double &fwd, double &rev) {
    fwd = 0.0;
    rev = 0.0;
    const uint32_t skip = nw | ~low_mask(W);                  // bit c: column c adds nothing
    const char *__restrict__ base = reinterpret_cast<const char *>(tab2);
    int c1 = 0;
    for (; W - c1 > 4; c1 += 8)                                // eight columns a step while more than four are left ...
        score_columns<8>(base, tab_bytes + (uint32_t) c1 * 64u, zero_bytes, (uint32_t) (cw >> (2 * c1)), skip >> c1, fwd, rev);
    if (W - c1 > 0)                                            // ... then four (loads are what the stage pays for)
        score_columns<4>(base, tab_bytes + (uint32_t) c1 * 64u, zero_bytes, (uint32_t) (cw >> (2 * c1)), skip >> c1, fwd, rev);
}

}  // namespace ms
