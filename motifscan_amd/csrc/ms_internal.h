// ms_internal.h -- shared declarations of libmotifscan_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/motifscan_amd.h"

namespace ms {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define MS_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) {                                                            \
            ms::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                          __LINE__);                                                        \
            return (e__ == hipErrorOutOfMemory) ? MS_ERR_NOMEM : MS_ERR_RUNTIME;            \
        }                                                                                   \
    } while (0)

// Measurement / A-B switches (MS_PF_*, MS_HIT_COORD) are honoured only when MS_MEASURE=1 is set as well: a stray
// variable in the environment must never change what the product does (tests/ and tools/ opt in explicitly).
inline const char *measure_env(const char *name) {
    const char *g = getenv("MS_MEASURE");
    if (!g || g[0] != '1' || g[1] != 0) return nullptr;
    return getenv(name);
}

// ------------------------------------------------------------------ limits / packing --
constexpr int kMaxFastWidth = 32;      // one lane holds 32 bases (64 bits of 2-bit codes)
constexpr int kMaxGroups = 16;         // 2-mer groups per motif (ceil(32 / 2))
constexpr int kMaxMotifs = 65000;      // 14-bit table-group id in a candidate record, >= 4 motifs per group
constexpr int kPadWords = 8;           // zero words after the packed codes (window reads run past the end)

constexpr int64_t kMaxBases = (1LL << 34) - 64;   // 34-bit position field of a candidate record

// candidate record (one per lane and table group): [63:30] global base position, [29:16] table group,
// [15:0] flags: bit n = field n flagged (motif slot n >> 1; even n forward, odd n reverse)
__host__ __device__ inline uint64_t cand_pack(uint64_t g, uint32_t group, uint32_t flags) {
    return (g << 30) | ((uint64_t) group << 16) | flags;
}

// ------------------------------------------------------------------ pre-filter plan --
constexpr int kGroupSlots = 8;         // motif slots per table group (4 at 16-bit fields, 6 at 10-bit)
constexpr int kMaxClasses = 2 * kMaxGroups;

struct ClassDesc {
    int32_t G;          // 2-mer groups of every table group in the class
    int32_t n_groups;
    int32_t fb;         // field bits: 16 (4 motifs per 16-byte entry) or 10 (6 motifs)
    uint32_t base16;    // offset of the class's tables inside the tile, 16-byte units
    int32_t first_group;   // global index of the class's first table group
};

struct TileDesc {
    uint32_t table_off16;   // offset into the table buffer, in 16-byte units
    uint32_t table_len16;   // tile size in 16-byte units
    int32_t first_group;    // global index of the tile's first table group
    int32_t n_classes;
    ClassDesc cls[kMaxClasses];
};

// Host-side result of planning: which motifs take the integer pre-filter, their quantised
// 2-mer tables in groups that share 16-byte entries, and how groups are cut into LDS tiles.
struct PrefilterPlan {
    int strand_mask = 0;
    int engine = 0;                      // 0: packed 2-mer tables read per lane from LDS; 1: int8 one-hot MFMA; 2: int8 Walsh-form MFMA; 3: fp6 x fp4 one-hot MFMA (below)
    std::vector<int32_t> fast_motifs;    // motif ids on the pre-filter path, in group order
    std::vector<int32_t> exact_motifs;   // motif ids scored in fp64 at every window
    std::vector<int32_t> group_motifs;   // [n_groups][kGroupSlots], -1 = empty slot
    std::vector<int32_t> group_G;        // [n_groups]
    std::vector<int32_t> group_fb;       // [n_groups]
    std::vector<uint32_t> tables;        // per group: [G][16 codes][4 words]; field n of motif slot j: n = 2j (fwd), 2j+1 (rev)
    std::vector<TileDesc> tiles;
    int64_t lds_bytes_per_position = 0;  // sum over groups of G * 16 bytes (per lane, per position)
};

// ---- engine 1: the pre-filter as an int8 matrix product on the matrix cores -------------------
// v_mfma_i32_32x32x32_i8: D[row][col] += sum_k A[row][k] * B[k][col].  Rows are (motif, strand)
// fields, columns are 32 consecutive window starts, and k runs over (column of the motif, base):
// B is the one-hot image of the sequence (1 where the base at window start + column is that base),
// A holds the quantised PWM entries.  One instruction covers 8 motif columns ("k-block"); a row
// tile (32 rows = 2 table groups of 8 motifs x {fwd, rev}) needs ceil(W_max / 8) of them.
//
// Operand bytes: lane l = 32 * khalf + r holds 16 bytes; byte i of k-block kb is motif column
// 8 * kb + 4 * khalf + (i >> 2), base i & 3 -- for A (r = row) and for B (r = window) alike, so the
// hardware's own k order never matters.  A row tile is stored as [kb][lane][16 bytes]: a wave reads
// its A operand with one conflict-free ds_read_b128 at lane * 16.
//
// Result register j of lane (r, khalf = h) is row (j & 3) + 8 * (j >> 2) + 4 * h.  Lane half h of
// row tile t therefore owns table group 2t + h, and register j is that group's field 15 - j
// (field n: motif slot n >> 1, even n forward, odd n reverse), so that shifting the 16 sign bits
// together in register order yields the flag word of a candidate record directly.
constexpr int kMfmaRowTileBytesPerKb = 1024;
inline int mfma_row_of(int h, int field) { const int j = 15 - field; return (j & 3) + 8 * (j >> 2) + 4 * h; }
inline size_t mfma_byte_index(int kb, int row, int col_in_kb, int base) {     // inside a row tile
    const int khalf = col_in_kb >> 2;
    return (size_t) kb * kMfmaRowTileBytesPerKb + (size_t) (khalf * 32 + row) * 16 + (size_t) (col_in_kb & 3) * 4 + base;
}

// ---- engine 2: the same product with THREE k-slots per base instead of four ---------------------
// A base has four states, so any per-column score table is  c0 + c1*s1 + c2*s2 + c3*s1*s2  with
// s1 = +1/-1 by bit 0 of the base code and s2 = +1/-1 by bit 1 (Walsh form): three k-slots per column, the
// constants c0 summed into a per-row bias.  32 k-slots = 10 columns (5 per lane half: bytes 3j..3j+2 of the
// half hold c1, c2, c3 of its j-th column) + one spare byte per half; the B operand carries (s1, s2, s1*s2) per
// base and the constants 64 (half 0) / 1 (half 1) in the spare bytes, so the A-side spare bytes of k-block 0
// give the row bias 64 * a_hi + a_lo.  Rows tiles need ceil(W / 10) k-blocks: 18 % fewer matrix instructions on
// the JASPAR width distribution than engine 1's 8 columns per k-block.
constexpr int kW2Cols = 10;            // motif columns per k-block
constexpr int kW2MaxWidth = 30;        // 3 k-blocks
inline size_t mfma2_byte_index(int kb, int row, int col_in_kb, int slot) {      // slot 0..2 = c1, c2, c3; inside a row tile
    const int khalf = col_in_kb / 5;
    return (size_t) kb * kMfmaRowTileBytesPerKb + (size_t) (khalf * 32 + row) * 16 + (size_t) (col_in_kb % 5) * 3 + slot;
}
inline size_t mfma2_spare_index(int row, int khalf) {                           // k-block 0
    return (size_t) (khalf * 32 + row) * 16 + 15;
}

// ---- engine 3: the product on the FP6 x FP4 block-scaled matrix instruction ---------------------------------------------
// v_mfma_scale_f32_32x32x64_f8f6f4 takes K = 64 per instruction in the time the int8 instruction takes for K = 32 (measured:
// profiles/r02_mfma_f6_probe.log), i.e. SIXTEEN motif columns per k-block.  A (PWM side) is fp6 e2m3, B (sequence side) the
// one-hot image in fp4 e2m1 (1.0 = code 0x2), both with block scale 127 = 2^0; the f32 result is exact (entries are multiples
// of 1/8 of magnitude <= 7.5, at most 32 of them add up).  Operand layout, probed with exact data: lane l = 32 * khalf + r
// holds row (A) / window (B) r and the 32 consecutive k = 32 * khalf + j, value j at bits [6j, 6j + 6) of the lane's 192 bits
// (A) / bits [4j, 4j + 4) of its 128 bits (B); here k-slot j of lane half khalf = motif column 8 * khalf + (j >> 2), base j & 3.
// A row tile is stored per k-block as [plane 0..2][lane 0..63][8 bytes] (plane p = bits [64p, 64p + 64) of the lane's field):
// three conflict-free ds_read_b64 at lane * 8.
// Entries are in units of 1/8: v_c(b) = t_c - dq_c(b) with dq on the e2m3 grid {0..16, 18..32 step 2, 36..60 step 4}, 56 budget
// levels, a clamped deficit = 60, and the offsets t = 16, 16, 16, 8 on the first four columns (multiples of 4 keep every
// difference on the grid), so that acc = (56 - sum dq) / 8 >= 0  <=>  candidate: the sign bit of the f32 result.
constexpr int kF6Cols = 16;                     // motif columns per k-block
constexpr int kF6BytesPerKb = 3 * 64 * 8;       // 1536
constexpr int kF6Levels = 56;
inline bool f6_representable(int u) {           // |u| in units of 1/8
    const int m = u < 0 ? -u : u;
    return m <= 16 || (m <= 32 && (m & 1) == 0) || (m <= 60 && (m & 3) == 0);
}
inline uint32_t f6_code(int u) {                // e2m3: sign | exp(2) | mant(3), bias 1; u must be representable
    const uint32_t sgn = u < 0 ? 32u : 0u;
    const int m = u < 0 ? -u : u;
    if (m == 0) return 0u;                      // never -0
    if (m < 8) return sgn | (uint32_t) m;
    if (m < 16) return sgn | (1u << 3) | (uint32_t) (m - 8);
    if (m < 32) return sgn | (2u << 3) | (uint32_t) (m / 2 - 8);
    return sgn | (3u << 3) | (uint32_t) (m / 4 - 8);
}
inline int f6_value(uint32_t code) {            // back to units of 1/8
    const int e = (int) (code >> 3) & 3, m = (int) code & 7;
    const int v = e == 0 ? m : (8 + m) << (e - 1);
    return (code & 32u) ? -v : v;
}
inline size_t f6_bit_index(int kb, int row, int col_in_kb, int base, int *bit_in_lane) {   // byte offset of the lane's plane 0 word inside a row tile
    const int khalf = col_in_kb >> 3;
    *bit_in_lane = 6 * ((col_in_kb & 7) * 4 + base);
    return (size_t) kb * kF6BytesPerKb + (size_t) (khalf * 32 + row) * 8;
}
inline void f6_put(uint8_t *tile, int kb, int row, int col_in_kb, int base, uint32_t code) {
    int bit;
    const size_t lane_off = f6_bit_index(kb, row, col_in_kb, base, &bit);
    for (int i = 0; i < 6; i++) {
        const int bb = bit + i;                 // plane bb / 64, bit bb % 64 of that plane's 8-byte word
        uint8_t *byte = tile + lane_off + (size_t) (bb >> 6) * 512 + (size_t) ((bb & 63) >> 3);
        if ((code >> i) & 1u) *byte |= (uint8_t) (1u << (bb & 7)); else *byte &= (uint8_t) ~(1u << (bb & 7));
    }
}
inline uint32_t f6_get(const uint8_t *tile, int kb, int row, int col_in_kb, int base) {
    int bit;
    const size_t lane_off = f6_bit_index(kb, row, col_in_kb, base, &bit);
    uint32_t code = 0;
    for (int i = 0; i < 6; i++) {
        const int bb = bit + i;
        const uint8_t byte = tile[lane_off + (size_t) (bb >> 6) * 512 + (size_t) ((bb & 63) >> 3)];
        code |= (uint32_t) ((byte >> (bb & 7)) & 1u) << i;
    }
    return code;
}

// Quantiser + planner (pure host code, ms_plan.cpp).  lds_budget in bytes; min_field_bits 10 or 16.
// engine 1 (build_plan_mfma): ClassDesc.G = k-blocks per row tile, .n_groups = ROW TILES in the class,
// .first_group = table group of its first row tile; group_G = k-blocks, group_fb = 8.
int build_plan_mfma(const double *values, const int64_t *val_off, const int32_t *widths,
                    const double *cutoffs, const double *max_raw, int32_t n_pwms, int strand_mask,
                    size_t lds_budget, int engine, PrefilterPlan *plan);
int build_plan(const double *values, const int64_t *val_off, const int32_t *widths,
               const double *cutoffs, const double *max_raw, int32_t n_pwms, int strand_mask,
               size_t lds_budget, int min_field_bits, PrefilterPlan *plan);

// Sort (ms_sort.hip): keys ascending over bits [0, end_bit).  Query temp size with temp == nullptr.
int sort_hit_pairs(void *temp, size_t *temp_bytes, const uint64_t *keys_in, uint64_t *keys_out,
                   const double *vals_in, double *vals_out, size_t n, int end_bit, hipStream_t stream);

// Keys-only sort of the candidate entries (motif | coordinate) over bits [0, end_bit).
int sort_keys(void *temp, size_t *temp_bytes, const uint64_t *keys_in, uint64_t *keys_out, size_t n, int end_bit, hipStream_t stream);

// Descending sort of one row of fp64 scores (cutoff builder).  Query temp size with temp == nullptr.
int sort_doubles_desc(void *temp, size_t *temp_bytes, const double *in, double *out, size_t n, hipStream_t stream);

// Exclusive prefix sum of 32-bit counts into 64-bit slots (destinations of the hits that survive de-duplication, of the
// sites a sweep hands out: more than 2^32 of them per call are legal).
int exclusive_sum_u32(void *temp, size_t *temp_bytes, const uint32_t *in, uint64_t *out, size_t n, hipStream_t stream);

}  // namespace ms
