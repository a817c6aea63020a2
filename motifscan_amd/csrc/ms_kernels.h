// ms_kernels.h -- kernel argument structs and launchers (ms_kernels.hip).
#pragma once
#include "ms_internal.h"

namespace ms {

constexpr int kPfThreads = 1024;      // pre-filter block: 16 waves, one block per CU (4 waves per SIMD: <= 128 VGPRs) -- its waves never meet at a barrier
constexpr int kPfBlocksPerCu = 1;     // after the tables are loaded, so the block shape only decides how many copies of the tables a CU's LDS holds
constexpr int kPfCounters = 8;        // counter words per LDS tile of the per-wave hand-out (64 bytes apart)
constexpr size_t kPfClsBytes = (size_t) kMaxClasses * 32;   // the tile's class descriptors, 32 bytes apiece, behind the B-operand table (prefilter_f6_kernel)
constexpr size_t kF6LutBytes = 256 * 8 + kPfClsBytes;       // byte of four 2-bit codes -> 16 fp4 one-hot k-slots (8 bytes), after the tile's tables; + the class descriptors
constexpr int kPfStageWords = 24;            // per wave: the current pass's code words, then its non-ACGT words -- 8 + 4 (64 window starts), or 16 + 8 (the double pass: 128), after the B-operand table
constexpr size_t kPfStageBytes = (size_t) (kPfThreads / 64) * kPfStageWords * sizeof(uint32_t);
// per wave: the lanes that hold a candidate park their 16 result registers here; the flag words are decoded later, one parked entry
// per lane (ms_kernels.hip, "candidate hand-off")
constexpr int kPfClkWords = 2 + kMaxClasses;
constexpr int kRareCapMin = 16;              // entries per wave: whatever LDS the tile's tables leave, between these two (PfArgs::rare_cap)
constexpr int kRareCapMax = 64;              // (a wave has 64 lanes: an event never needs more)
constexpr int kRareEntryWords = 12;          // the flag bytes of the 16 result registers (8 words for paired rows, 4 for plain ones) + {position low word, position high bits | group << 8 | paired << 31} at byte 32 + 2 spare: 48 bytes (ms_kernels.hip, park_store)
constexpr size_t kPfRareBytesMin = (size_t) (kPfThreads / 64) * kRareCapMin * kRareEntryWords * sizeof(uint32_t);
constexpr size_t kPfOnehotBytes = (size_t) (kPfThreads / 64) * 160 * 16;  // per wave: the double pass's one-hot array, 160 entries of 16 bytes (ms_kernels.hip, kOnehotEntries)
constexpr double kDenseHitsPerHalfTile = 12.0; // the dense-candidate kernel (ms_kernels.hip) above this many expected hits per row tile and 64 windows: measured 4.4 (p = 1e-3) 34.9 against
                                               // 31.9 ms parked, 44 (p = 1e-2, 1/8 shard) 5.7 against 9.5 ms (profiles/r05_dense_form.log)
constexpr int kPfEmitWords = 16;             // per wave: its place in the global candidate list and the launch's constants (PfEmit, ms_kernels.hip)
constexpr size_t kPfEmitBytes = (size_t) (kPfThreads / 64) * kPfEmitWords * sizeof(uint32_t);

struct DevSeq {
    const uint32_t *codes;
    const uint32_t *nmask;
    const int64_t *offsets;   // [R+1]
    const int32_t *blk2reg;   // [n_bases/64 + 2] region of position 64*b
    const int4 *blkinfo;      // [n_bases/64 + 2] {that region, its start, the next region's, the one after's: relative to 64*b} (blk2reg_kernel)
    int64_t R;
    int64_t n_bases;
};

struct DevPwm {
    const double2 *tab2;      // see ms_kernels.hip header
    const int64_t *tab_off;   // [P] offset of motif p in tab2 (double2 units)
    const int32_t *width;     // [P]
    const double *max_raw;    // [P]
    const double *cutoff;     // [P]
    const double *raw_floor;  // [P] a raw (un-normalised) sum below this can never pass the hit test (-inf if unknown)
    int32_t P;
    uint32_t zero_bytes;      // byte offset of tab2's closing all-zero entry (what a column that adds nothing reads: score_window32)
    int32_t tab32;            // != 0: every entry of tab2 lies below 4 GB, byte offsets fit 32 bits
    const double *thresh;     // [P][4] {max_raw, cutoff, raw_floor, 0}: what the hit test reads, side by side
};

// What rescore_kernel needs to know about a field of a table group, in one 16-byte read (plan-specific: ms_pwmset::d_field_meta)
struct FieldMeta {
    int32_t motif;            // -1: empty field
    int32_t width;
    uint32_t tab_bytes;       // byte offset of the motif's table in DevPwm::tab2 (DevPwm::tab32)
    float floor32;            // the motif's raw-sum floor (DevPwm::raw_floor) rounded DOWN to a float: a sum below it is no hit
};

struct HitOut {
    uint64_t *keys;           // motif << (gbits+1) | coordinate << 1 | strand bit; coordinate = (region << pbits | position in
                              // the region) when pbits > 0 (gbits = rbits + pbits), else the global base position
    double *vals;             // normalised fp64 score
    unsigned long long *n_hits;
    uint64_t cap;
    int gbits;                // bits of the coordinate field
    int pbits;                // > 0: bits of the position inside a region (see keys)
};

struct PfArgs {
    const uint32_t *codes;
    const uint32_t *nmask;
    int64_t n_bases;
    const uint4 *tables;
    const TileDesc *tiles;
    uint32_t lut_off16;       // start of the B-operand table in dynamic LDS (16-byte units)
    uint32_t stage_off16;     // start of the per-wave sequence staging
    uint32_t rare_off16;      // start of the per-wave parking space for candidate lanes' result registers
    uint32_t emit_off16;      // start of the per-wave PfEmit
    uint32_t onehot_off16;    // start of the per-wave one-hot arrays
    uint32_t rare_cap;        // entries of a wave's parking space; decoded when fewer than 8 (above 32 entries: a quarter) are free
    uint64_t *cand;           // candidate records; a wave reserves blocks of cand_block slots (unused slots are written as 0 = empty)
    unsigned long long *n_cand;   // slots reserved so far
    uint64_t cand_cap;
    uint32_t cand_block;      // >= 64
    uint64_t cand_static;     // slots [0, cand_static) are the launch's waves' own first blocks (wave w: [w, w + 1) * cand_block)
    int skip_alln;            // != 0: no motif of the plan reports a window made of non-ACGT bases only: such windows are dropped unseen
    int no_emit;              // measurement only (MEAS instantiations): run the filter, drop the candidates
    int cls_clk;              // measurement only: also time the classes (MS_PF_CLOCK=2: the stamps themselves cost a few per cent)
    unsigned long long *clk;  // measurement only: per block kPfClkWords words {shader cycles, 100 MHz ticks, wave 0's cycles inside each class}, or nullptr
    unsigned int *chunk_counter;   // [LDS tiles][kPfCounters] words 64 bytes apart, zeroed: the units behind the waves' own first ones
    int use_counters;              // 0: a small input, one even unit per wave and no atomics
    int wave_passes;               // per-wave hand-out: passes of 64 window starts a wave takes per atomic (scan_locked sizes it)
};


int launch_pack(const uint8_t *ascii, int64_t n_bases, uint32_t *codes, uint32_t *nmask, hipStream_t st);
int prefilter_set_lds(bool wide, bool meas, bool dense, size_t bytes, int floor_ = 0);
int launch_prefilter(const PfArgs &A, bool wide, bool meas, bool dense, int blocks_per_tile, int n_tiles, size_t lds_bytes, hipStream_t st, int floor_ = 0);
int launch_exact_all(const DevSeq &S, const DevPwm &Pw, const int32_t *motifs, int32_t n_motifs, int strand_mask,
                     const HitOut &H, hipStream_t st, int max_width);
// the same for long lists: chunks of the list in motif order, the window carried along (rescore_carry_kernel); one 1024-thread block per CU
int rescore_carry_set_lds();
int launch_rescore_carry(const DevSeq &S, const DevPwm &Pw, const uint64_t *cand, const unsigned long long *n_cand, uint64_t n_static,
                         uint64_t cand_cap, const FieldMeta *field_meta, int strand_mask, const HitOut &H, int n_blocks, hipStream_t st);
int launch_rescore(const DevSeq &S, const DevPwm &Pw, const uint64_t *cand, const unsigned long long *n_cand, uint64_t n_static,
                   uint64_t cand_cap, const FieldMeta *field_meta, int strand_mask, const HitOut &H, int n_blocks,
                   hipStream_t st);
// after a sort over the key bits above kSortLowBits: runs of hits equal in those bits into full key order (in place)
int launch_sort_fixup(uint64_t *keys, double *vals, int64_t n, const unsigned long long *n_dev, hipStream_t st);
int launch_finalize(const uint64_t *keys, int64_t n, const unsigned long long *n_dev, int gbits, int rbits, int pbits, int32_t P, const DevSeq &S,
                    int64_t *seq_idx, int64_t *pos, int8_t *strand, int64_t *motif_first, unsigned long long *region_counts,
                    hipStream_t st);
int launch_fill_tail(uint64_t *keys, const unsigned long long *n_dev, uint64_t cap, hipStream_t st);
int launch_extract(const uint32_t *gcodes, const uint32_t *gnmask, const int64_t *src_start, const int64_t *dst_off,
                   int64_t R, int64_t n_out, uint32_t *codes, uint32_t *nmask, hipStream_t st);
int launch_blk2reg(const int64_t *offsets, int64_t R, int64_t n_bases, int32_t *blk2reg, int4 *blkinfo, hipStream_t st);
int launch_score(const DevSeq &S, const DevPwm &Pw, int strand_mask, double *out, hipStream_t st);
int launch_sweep_count(int64_t n, const int64_t *motif_off, int32_t P, const int32_t *width, const int64_t *pos,
                       int32_t window, int32_t stride, int64_t n_windows, uint32_t *cnt, hipStream_t st);
// counts only, from the unordered hit keys (ms_regions.hip)
bool count_only_supported(int32_t P, int64_t R, int pbits);
size_t count_only_bitmap_words(int32_t P, int64_t R);
int launch_count_only(const uint64_t *keys, int64_t n, const unsigned long long *n_dev, int gbits, int pbits, int64_t R, int32_t P, uint32_t *bitmap,
                      unsigned long long *region_counts, unsigned long long *motif_hits, int64_t *motif_first, hipStream_t st);
int launch_sweep_countonly(int64_t n, const int64_t *motif_off, int32_t P, const int32_t *width, const int64_t *pos,
                           int32_t window, int32_t stride, int64_t n_windows, unsigned long long *region_counts, unsigned long long *n_sites, hipStream_t st);
int launch_sweep_scatter(int64_t n, const int64_t *motif_off, int32_t P, const int32_t *width, const int64_t *pos,
                         const double *score, const int8_t *strand, const uint64_t *dst, int32_t window, int32_t stride,
                         int64_t n_windows, int64_t total, int64_t *seq_idx_out, int64_t *pos_out, double *score_out,
                         int8_t *strand_out, int64_t *motif_off_out, unsigned long long *region_counts, hipStream_t st);
int launch_dedup(int64_t n, const int64_t *motif_off, int32_t P, const int32_t *width, const int64_t *seq_idx,
                 const int64_t *pos, const double *score, const int8_t *strand, uint32_t *keep, hipStream_t st);
int launch_compact_hits(int64_t n, const uint32_t *keep, const uint64_t *dst, const int64_t *seq_in, const int64_t *pos_in,
                        const double *score_in, const int8_t *strand_in, int64_t *seq_out, int64_t *pos_out,
                        double *score_out, int8_t *strand_out, const int64_t *off_in, int32_t P, int64_t *off_out,
                        hipStream_t st);
int launch_site_tables(int64_t n, const int64_t *motif_off, int32_t P, int64_t R, const int64_t *seq_idx,
                       const double *score, int32_t *n_sites, double *max_score, hipStream_t st);
int launch_pack_hits(int64_t n, const unsigned long long *n_dev, const int64_t *seq_idx, const int64_t *pos, const int8_t *strand, uint64_t *coord,
                     unsigned int *bad, hipStream_t st, int shift = 0);
int launch_gather_ranks(const double *sorted, int64_t n, const int64_t *ranks, int32_t n_ranks, double *out, hipStream_t st);

}  // namespace ms
