// ms_internal.h -- shared declarations of libmotifscan_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
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

// ------------------------------------------------------------------ limits / packing --
constexpr int kMaxFastWidth = 32;      // one lane holds 32 bases (64 bits of 2-bit codes)
constexpr int kMaxGroups = 16;         // 2-mer groups per motif (ceil(32 / 2))
constexpr int kMaxMotifs = 65535;      // 16-bit motif id in candidate / hit keys
constexpr int kPadWords = 8;           // zero words after the packed codes (window reads run past the end)

constexpr int64_t kMaxBases = (1LL << 34) - 64;   // 34-bit position field of a candidate record

// candidate record (one per lane and pair of quads): [63:30] global base position, [29:16] first quad,
// [15:0] flags: bit slot (0..7, over the two quads) = forward field flagged, bit 8 + slot = reverse
__host__ __device__ inline uint64_t cand_pack(uint64_t g, uint32_t quad, uint32_t flags) {
    return (g << 30) | ((uint64_t) quad << 16) | flags;
}

// ------------------------------------------------------------------ pre-filter plan --
struct ClassDesc {
    int32_t G;         // 2-mer groups of every quad in the class
    int32_t n_quads;
};

struct TileDesc {
    uint32_t table_off16;   // offset into the table buffer, in 16-byte units
    uint32_t table_len16;   // tile size in 16-byte units
    int32_t first_quad;     // global index of the tile's first quad
    int32_t n_classes;
    ClassDesc cls[kMaxGroups];
};

// Host-side result of planning: which motifs take the integer pre-filter, their quantised
// 2-mer tables grouped in quads of four motifs, and how quads are cut into LDS tiles.
struct PrefilterPlan {
    int strand_mask = 0;
    std::vector<int32_t> fast_motifs;    // motif ids on the pre-filter path, in quad order
    std::vector<int32_t> exact_motifs;   // motif ids scored in fp64 at every window
    std::vector<int32_t> quad_motifs;    // [n_quads][4], -1 = empty slot
    std::vector<int32_t> quad_G;         // [n_quads]
    std::vector<uint32_t> tables;        // per quad: [G][16 codes][4 slots] words (lo16 fwd, hi16 rev)
    std::vector<TileDesc> tiles;
    int64_t lds_bytes_per_position = 0;  // sum over quads of G * 16 bytes (per lane, per position)
};

// Quantiser + planner (pure host code, ms_plan.cpp).  lds_budget in bytes.
int build_plan(const double *values, const int64_t *val_off, const int32_t *widths,
               const double *cutoffs, const double *max_raw, int32_t n_pwms, int strand_mask,
               size_t lds_budget, PrefilterPlan *plan);

// Sort (ms_sort.hip): keys ascending over bits [0, end_bit).  Query temp size with temp == nullptr.
int sort_hit_pairs(void *temp, size_t *temp_bytes, const uint64_t *keys_in, uint64_t *keys_out,
                   const double *vals_in, double *vals_out, size_t n, int end_bit, hipStream_t stream);

}  // namespace ms
