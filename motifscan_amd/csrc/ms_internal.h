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

// Measurement switches (MS_PF_NOEMIT, MS_PF_CLOCK, MS_PF_MAX_BLOCKS, MS_SORT_FULL, MS_SORT_FIXUP_MIN, MS_RESCORE_SORTED_MIN, MS_BLKINFO_FAR, MS_HIT_COORD, MS_CU_PARTITION) are honoured only when
// MS_MEASURE=1 is set as well: a stray variable in the environment must never change what the product does (tests/ and tools/
// opt in explicitly).
inline const char *measure_env(const char *name) {
    const char *g = getenv("MS_MEASURE");
    if (!g || g[0] != '1' || g[1] != 0) return nullptr;
    return getenv(name);
}

// ------------------------------------------------------------------ limits / packing --
constexpr int kMaxFastWidth = 63;      // widest motif the pre-filter takes: 4 k-blocks of 16 columns, the last column carries the row bias
constexpr int kMaxMotifs = 65000;      // 14-bit table-group id in a candidate record, >= 8 motifs per group
constexpr int kPadWords = 16;          // zero words after the packed codes (window reads run past the end)

constexpr int64_t kMaxBases = (1LL << 34) - 128;   // 34-bit position field of a candidate record

// candidate record (one per lane and table group): [63:30] global base position, [29:16] table group,
// [15:0] flags: bit n = field n of the group flagged (PrefilterPlan::group_fields)
__host__ __device__ inline uint64_t cand_pack(uint64_t g, uint32_t group, uint32_t flags) {
    return (g << 30) | ((uint64_t) group << 16) | flags;
}

// ------------------------------------------------------------------ pre-filter plan --
//
// The pre-filter is a rigorous UPPER bound of both strand scores of every window, evaluated as a matrix product on the matrix
// cores: v_mfma_scale_f32_32x32x64_f8f6f4 with A = the quantised PWM entries in fp6 (e2m3), B = the one-hot image of the
// sequence in fp4 (e2m1, 1.0 = code 0x2), both block scales 2^0; K = 64 per instruction = SIXTEEN motif columns x 4 bases
// ("k-block").  The f32 result is exact (entries are multiples of 1/8 of magnitude <= 7.5, at most 64 of them add up).
//
// Rows are (motif, strand) "fields", columns are 32 consecutive window starts.  A ROW TILE = the 32 rows of one instruction =
// two TABLE GROUPS of 16 fields (the unit a candidate record names): with both strands scanned a group holds 8 motifs x
// {fwd, rev} (field n: motif slot n >> 1, even n forward, odd n reverse), with one strand 16 motifs (field n: slot n).
// Result register j of lane (r, khalf = h) is row (j & 3) + 8 * (j >> 2) + 4 * h, so lane half h of row tile t owns table group
// 2t + h, and register j is that group's field 15 - j: shifting the 16 sign bits together in register order yields the flag
// word of a candidate record.
//
// Operand layout (probed with exact data, profiles/r02_mfma_f6_probe.log): lane l = 32 * khalf + r holds row (A) / window (B) r
// and the 32 consecutive k = 32 * khalf + j, value j at bits [6j, 6j + 6) of the lane's 192 bits (A) / bits [4j, 4j + 4) of its
// 128 bits (B); k-slot j of lane half khalf = motif column 8 * khalf + (j >> 2), base j & 3.  A row tile is stored per k-block as
// [plane 0..2][lane 0..63][8 bytes] (plane p = bits [64p, 64p + 64) of the lane's field): three conflict-free ds_read_b64.
//
// Entries, in units of 1/8 (ms_plan.cpp): v_c(b) = t_c - dq_c(b), dq = the column's deficit against its best base, quantised
// DOWN onto the e2m3 grid {0..16, 18..32 step 2, 36..60 step 4}; 56 budget levels; deficits of 60 and more are stored as 60.
// The offsets t_c (multiples of 4, <= 16, and never more than the column's own best contribution) sit on the columns themselves;
// what is left of the budget, b0 = 56 - sum t_c, is the ROW BIAS, stored for all four bases in the LAST column of the row tile's
// last k-block (column 16 * kb - 1: a motif of width W takes kb = W / 16 + 1 k-blocks).  acc = b0 + sum_c (t_c - dq_c) =
// (56 - sum dq) / 8 >= 0  <=>  candidate: the sign bit of the f32 result.
// Non-ACGT bases are all-zero one-hot columns (the kernel clears them in the B operand; the bias column is exempt), exactly the
// reference's "adds nothing" (cscore.c:345-353): because t_c never exceeds the column's best contribution, dropping a column
// can only lower acc by less than the true score loses, so windows with N go through the same filter -- no separate N path.
//
// PAIRED ROWS (round 3; motifs of <= 15 columns, i.e. most of a JASPAR-like set): the A operand's block scale is per lane = per
// (row, k-half), so one matrix row can carry TWO fields.  k-half 0 holds 8 columns of field X, k-half 1 holds 8 columns of field Y,
// BOTH k-halves of the B operand hold the same 8 bases of the window, the block scales are 2^-6 (X) and 2^-18 (Y), and the
// accumulator starts at the inline constant 4.0 -- whose unit in the last place, 2^-21, is then exactly one level (1/8) of field Y and
// 2^-12 of a level of field X.  With both fields' sums offset by 1024 levels the result is
//     4.0 + 2^-21 (2^12 (X + 1024) + (Y + 1024)),     X, Y = the two fields' sums in levels, -1024 <= X, Y < 1024:
// exact in f32 (an integer below 2^23 on top of the fixed leading bit), bit pattern 0x40800000 | (X + 1024) << 12 | (Y + 1024), i.e.
// bit 22 <=> X >= 0, bit 10 <=> Y >= 0 (kPairMask).  The offset rides on the bias column: its four k-slots of the B operand are
// not one-hot but the constants (6, 6, 6, 1) (kPairBiasB), and the field's four bias entries u_j on the e2m3 grid satisfy
// 6 (u_0 + u_1 + u_2) + u_3 = b0 + 1024 (pair_bias_entries: always solvable for |b0| <= 60).  A paired row tile is 32 rows x {X, Y}
// = FOUR table groups (lane half h, field select s: group 4t + 2h + s) of W / 8 + 1 HALF-blocks (8 columns, the last column of the
// last one is the bias column), so a result register answers for two (motif, strand) rows: half the result inspection per motif,
// and motifs of <= 7 columns cost half the matrix instructions.  |entry| <= 60 and <= 15 columns keep |X|, |Y| <= 960.
// Probed with exact data over the whole range, chained instructions included: tools/ubench/pair_probe.hip,
// profiles/r03b_pair_probe.log.
constexpr int kF6Cols = 16;                     // motif columns per k-block (the last one of a row tile is the bias column)
constexpr int kF6MaxKb = 4;
constexpr int kF6BytesPerKb = 3 * 64 * 8;       // 1536
constexpr int kF6Levels = 56;
constexpr int kGroupFields = 16;                // fields per table group = result registers per lane
constexpr int kMaxClasses = 6;                  // paired 1 / 2 half-blocks, plain 1 ... 4 k-blocks
constexpr int kPairCols = 8;                    // columns per half-block of a paired row
constexpr int kPairMaxWidth = 2 * kPairCols - 1;
// Motifs of 16 ... 23 columns COULD ride paired rows of three half-blocks at 36 budget levels (deficits of 40 and more stored as 40:
// 24 x 40 + 60 < 1024; set kPairWideMaxWidth = 3 * kPairCols - 1 and dispatch f6_pair_class<3>): built and measured -- 38 instead of
// 41 instructions per 32 windows on the benchmark set, pre-filter 18.9-19.2 against 19.3 ms per 500 Mbase, but the coarser rows pass
// 4 % more candidates and the fp64 stage takes back what the pre-filter saved (3.4 against 3.2 ms; profiles/r03k_ab_full.log).  Off.
constexpr int kPairWideMaxWidth = kPairMaxWidth;
constexpr int kPairWideLevels = 36;
constexpr int kPairScaleX = 127 - 6;            // E8M0 block scales of the two k-halves: X one level = 2^-9, Y one level = 2^-21
constexpr int kPairScaleY = 127 - 18;
constexpr float kPairC = 4.0f;                  // inline constant of the matrix instruction; ulp(4.0) = 2^-21
constexpr int kPairOffset = 1024;               // added to both fields' sums (through the bias column)
constexpr uint32_t kPairPattern = 0x40800000u;  // result = kPairPattern | (X + 1024) << 12 | (Y + 1024)
constexpr uint32_t kPairMask = (1u << 22) | (1u << 10);
constexpr uint32_t kPairBiasB = 0x2777u;        // the bias column of the B operand: k-slots (6.0, 6.0, 6.0, 1.0) in fp4 (e2m1), low nibble first
constexpr int kPairBiasW[4] = {6, 6, 6, 1};

inline int f6_kb_of_width(int W) { return W / kF6Cols + 1; }
inline int pair_kb_of_width(int W) { return W / kPairCols + 1; }
inline int mfma_row_of(int h, int field) { const int j = 15 - field; return (j & 3) + 8 * (j >> 2) + 4 * h; }

struct ClassDesc {
    int32_t nk;            // matrix instructions per row tile and 32 windows: k-blocks (plain) / half-blocks (paired)
    int32_t n_row_tiles;
    uint32_t base16;       // offset of the class's tables inside the LDS tile, 16-byte units
    int32_t first_group;   // global index of the class's first table group
    int32_t paired;        // != 0: paired rows, four table groups per row tile
};

// where a table group's fields sit in the operand image (host side: tests decode the physical image through this)
struct GroupInfo {
    uint32_t tab_off;      // byte offset of the group's row tile in PrefilterPlan::tables
    int8_t nk;             // instructions per row tile
    int8_t paired;
    int8_t h;              // lane half of the group's result registers
    int8_t sel;            // paired rows: 0 = field X (k-half 0, scale 2^12), 1 = field Y
};

struct TileDesc {
    uint32_t table_off16;   // offset into the table buffer, in 16-byte units
    uint32_t table_len16;   // tile size in 16-byte units
    int32_t first_group;    // global index of the tile's first table group
    int32_t n_classes;
    int32_t max_nk;         // widest class of the tile
    ClassDesc cls[kMaxClasses];
};

// Host-side result of planning: which motifs take the pre-filter, their operand tables, and how row tiles are cut into LDS tiles.
struct PrefilterPlan {
    int strand_mask = 0;
    std::vector<int32_t> fast_motifs;    // motif ids on the pre-filter path, in group order
    std::vector<int32_t> exact_motifs;   // motif ids scored in fp64 at every window
    std::vector<int32_t> group_fields;   // [n_groups][kGroupFields] motif id of the field, -1 = empty
    std::vector<int32_t> group_kb;       // [n_groups] matrix instructions (k-blocks / half-blocks) of the group's row tile
    std::vector<int32_t> group_cols;     // [n_groups] columns of the group's fields, the bias column included (16 / 8 per instruction)
    std::vector<GroupInfo> group_info;   // [n_groups]
    std::vector<uint32_t> tables;        // the operand image, row tile after row tile
    std::vector<TileDesc> tiles;
    int64_t lds_bytes_per_position = 0;  // A-operand bytes read per window start
    int64_t kb_total = 0;                // k-blocks over all row tiles
    bool alln_can_hit = false;           // some pre-filter motif reports windows made of non-ACGT bases only (threshold <= 0)
};

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
// the four bias-column entries (units of 1/8, on the e2m3 grid) with 6 (u0 + u1 + u2) + u3 = total; false if there are none
inline bool pair_bias_entries(int total, int u[4]) {
    static const int grid[] = {60, 56, 52, 48, 44, 40, 36, 32, 30, 28, 26, 24, 22, 20, 18, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0};
    for (int a : grid)
        for (int b : grid) {
            if (b > a) continue;
            for (int c : grid) {
                if (c > b) continue;
                const int rest = total - 6 * (a + b + c);
                if (rest >= -60 && rest <= 60 && f6_representable(rest)) { u[0] = a; u[1] = b; u[2] = c; u[3] = rest; return true; }
            }
        }
    return false;
}
// Byte offset, inside a row tile of nk blocks, of plane `plane` (0 ... 2) of the 8-byte word that `lane` reads for block kb.  Tiles of one
// block (and of 3 or 4) are plane-major -- [block][plane][lane]: a lane's three words lie 512 bytes apart --, tiles of TWO blocks
// lane-major, 48 bytes per lane (the lane's three words of block 0, then of block 1): the kernel reads them with three 16-byte
// reads at the LDS array's full rate (ds_read_b128: 4 cycles; the strided pair read ds_read2st64_b64 takes 8 for the same bytes).
inline size_t f6_word_off(int nk, int kb, int lane, int plane) {
    return nk == 2 ? (size_t) lane * 48 + (size_t) (kb * 3 + plane) * 8 : (size_t) kb * kF6BytesPerKb + (size_t) plane * 512 + (size_t) lane * 8;
}
inline int f6_bit_index(int row, int col_in_kb, int base, int *lane) {                      // bit of the code's LSB inside the lane's 192 bits of the block
    *lane = (col_in_kb >> 3) * 32 + row;
    return 6 * ((col_in_kb & 7) * 4 + base);
}
inline void f6_put(uint8_t *tile, int nk, int kb, int row, int col_in_kb, int base, uint32_t code) {
    int lane;
    const int bit = f6_bit_index(row, col_in_kb, base, &lane);
    for (int i = 0; i < 6; i++) {
        const int bb = bit + i;                 // plane bb / 64, bit bb % 64 of that plane's 8-byte word
        uint8_t *byte = tile + f6_word_off(nk, kb, lane, bb >> 6) + (size_t) ((bb & 63) >> 3);
        if ((code >> i) & 1u) *byte |= (uint8_t) (1u << (bb & 7)); else *byte &= (uint8_t) ~(1u << (bb & 7));
    }
}
inline uint32_t f6_get(const uint8_t *tile, int nk, int kb, int row, int col_in_kb, int base) {
    int lane;
    const int bit = f6_bit_index(row, col_in_kb, base, &lane);
    uint32_t code = 0;
    for (int i = 0; i < 6; i++) {
        const int bb = bit + i;
        const uint8_t byte = tile[f6_word_off(nk, kb, lane, bb >> 6) + (size_t) ((bb & 63) >> 3)];
        code |= (uint32_t) ((byte >> (bb & 7)) & 1u) << i;
    }
    return code;
}

// Quantiser + planner (pure host code, ms_plan.cpp).  lds_budget in bytes.
// pair_rows: motifs of <= kPairMaxWidth columns go to paired rows (the product default; false = measurement only).
int build_plan(const double *values, const int64_t *val_off, const int32_t *widths, const double *cutoffs,
               const double *max_raw, int32_t n_pwms, int strand_mask, size_t lds_budget, bool pair_rows, PrefilterPlan *plan);

// Sort (ms_sort.hip): keys ascending over bits [begin_bit, end_bit) (stable).  Query temp size with temp == nullptr.
constexpr int kSortLowBits = 8;        // a scan's hits are radix-sorted over the key bits above these; sort_fixup_kernel orders the rest
int sort_hit_pairs(void *temp, size_t *temp_bytes, const uint64_t *keys_in, uint64_t *keys_out,
                   const double *vals_in, double *vals_out, size_t n, int begin_bit, int end_bit, hipStream_t stream);

// Descending sort of one row of fp64 scores (cutoff builder).  Query temp size with temp == nullptr.
int sort_doubles_desc(void *temp, size_t *temp_bytes, const double *in, double *out, size_t n, hipStream_t stream);

// Exclusive prefix sum of 32-bit counts into 64-bit slots (destinations of the hits that survive de-duplication, of the
// sites a sweep hands out: more than 2^32 of them per call are legal).
int exclusive_sum_u32(void *temp, size_t *temp_bytes, const uint32_t *in, uint64_t *out, size_t n, hipStream_t stream);

}  // namespace ms
