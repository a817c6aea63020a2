// ms_kernels.hip -- the gfx950 kernels of the PWM scan path and their launchers.
//
// Data in HBM (all owned by ms_seqset / ms_pwmset, see ms_api.hip):
//   codes   uint32 words, 16 bases per word, base i at bits [2*(i%16), 2*(i%16)+1]; A0 C1 G2 T3,
//           non-ACGT stored as 0.  Regions are concatenated with no padding in between
//           (region r = bases [offsets[r], offsets[r+1])).  kPadWords zero words follow.
//   nmask   uint32 words, 32 bases per word, bit set = non-ACGT base (cscore.c:109-110 "-1")
//   offsets int64[R+1]
//   tab2    per motif W*4 double2: tab2[c*4+b] = { M[b][c], M[3-b][W-1-c] }  (forward entry and
//           the reverse-strand entry the reference adds at the same column step, cscore.c:348-352)
//
// Kernels:
//   pack_kernel       ASCII -> codes + nmask                     (cscore.c:81-114)
//   prefilter_f6_kernel  rigorous upper bound of both strand scores for EVERY window (with or without non-ACGT bases) as an
//                     fp6 x fp4 one-hot product on the matrix cores (v_mfma_scale_f32_32x32x64_f8f6f4); emits candidates
//   exact_tiled_kernel / exact_all_kernel  fp64 scoring of every window for motifs the pre-filter cannot take (the motif's table in LDS / in HBM)
//   rescore_kernel    fp64 scoring of the candidates, in the reference's order of operations,
//                     and the reference's hit test (cscore.c:356-358, 373-375); short candidate lists
//   rescore_carry_kernel   the same for long lists: chunks of the list in motif order, the windows read in list order and
//                     carried through an in-LDS counting sort (few cache lines per read either way)
//   sort_fixup_kernel the radix sort covers the key bits above the low eight; this orders the short runs that agree in them
//   finalize_kernel   sorted keys -> (seq_idx, pos, strand), per-motif offsets, region counts
//   score_kernel      c_score: first W bases of every sequence    (cscore.c:191-224)
//   gather_ranks_kernel   the rank pick of the cutoff builder    (motif/__init__.py:393-399)
//   dedup / compact / site_tables kernels   scanner.py:156-193 and io/__init__.py:23-33 on the sorted hits
//   extract_kernel    regions cut out of a genome that is resident as 2-bit codes (scanner.py:71-87)
//   blk2reg_kernel    region of every 64th position, so later position -> region lookups are O(1)
#include <algorithm>
#include <cstdlib>

#include "ms_device.h"

namespace ms {

// The middle field of a hit key: (region << pbits) | position inside the region when the set's regions are short
// enough for that to fit (H.pbits > 0: finalize then only unpacks bits), else the global base position.
__device__ __forceinline__ int64_t hit_coord(const HitOut &H, const DevSeq &S, int64_t r, int64_t g) {
    return H.pbits ? (int64_t) (((uint64_t) r << H.pbits) | (uint64_t) (g - S.offsets[r])) : g;
}

__device__ __forceinline__ void emit_hit(const HitOut &H, uint32_t motif, int64_t g, uint32_t sbit, double score) {
    const unsigned long long i = atomicAdd(H.n_hits, 1ULL);
    if (i < H.cap) {
        H.keys[i] = ((uint64_t) motif << (H.gbits + 1)) | ((uint64_t) g << 1) | sbit;
        H.vals[i] = score;
    }
}

// the reference's normalisation and threshold test, verbatim (cscore.c:356-358 / 373-375)
__device__ __forceinline__ void test_and_emit(const HitOut &H, const DevPwm &Pw, uint32_t motif, int64_t g,
                                              double fwd, double rev, int strand_mask) {
    // raw_floor: (cutoff - 1e-10) * max_raw minus 2000x the worst fp64 rounding of the sum, the divide and
    // the subtract (ms_api.hip): below it the reference's test is false whatever the roundings do, so the
    // two IEEE divides are only paid by windows that can actually be hits
    const double floor_ = Pw.raw_floor[motif];
    const bool try_f = (strand_mask & 1) && !(fwd < floor_);
    const bool try_r = (strand_mask & 2) && !(rev < floor_);
    if (!try_f && !try_r) return;
    const double max_raw = Pw.max_raw[motif];
    const double cutoff = Pw.cutoff[motif];
    const double s_f = try_f ? fwd / max_raw : 0.0, s_r = try_r ? rev / max_raw : 0.0;
    const bool hit_f = try_f && s_f - cutoff >= -1e-10, hit_r = try_r && s_r - cutoff >= -1e-10;
    if (hit_f) emit_hit(H, motif, g, 0u, s_f);
    if (hit_r) emit_hit(H, motif, g, 1u, s_r);
}

// Block-level staging of hits in LDS: one global atomicAdd per ~2000 hits instead of one per wave
// (all hit writers of the chip share ONE counter word; it sustains only ~90 M atomics/s).
constexpr int kHitStage = 2048;
template <int N>
struct HitStageN {
    static constexpr int kCap = N;
    uint64_t keys[N];
    double vals[N];
    unsigned int n;
    unsigned long long base;
};
typedef HitStageN<kHitStage> HitStage;

template <class ST>
__device__ __forceinline__ void stage_hit(ST &st, const HitOut &H, uint32_t motif, int64_t g, uint32_t sbit, double score) {
    const unsigned int i = atomicAdd(&st.n, 1u);
    if (i < (unsigned int) ST::kCap) {
        st.keys[i] = ((uint64_t) motif << (H.gbits + 1)) | ((uint64_t) g << 1) | sbit;
        st.vals[i] = score;
    } else {
        emit_hit(H, motif, g, sbit, score);            // stage full: straight to HBM
    }
}

// all threads of the block, at a block-uniform point
template <class ST>
__device__ __forceinline__ void stage_flush(ST &st, const HitOut &H) {
    __syncthreads();
    const unsigned int n = st.n < (unsigned int) ST::kCap ? st.n : (unsigned int) ST::kCap;
    if (threadIdx.x == 0 && n > 0) st.base = atomicAdd(H.n_hits, (unsigned long long) n);
    __syncthreads();
    const unsigned long long base = st.base;
    for (unsigned int i = threadIdx.x; i < n; i += blockDim.x)
        if (base + i < H.cap) { H.keys[base + i] = st.keys[i]; H.vals[base + i] = st.vals[i]; }
    __syncthreads();
    if (threadIdx.x == 0) st.n = 0;
    __syncthreads();
}

// floor32: FieldMeta::floor32, the raw-sum floor of test_and_emit rounded DOWN to a float (it came with the field's record: windows
// that cannot be hits read nothing more); the rest of the test reads {max_raw, cutoff} side by side
template <class ST>
__device__ __forceinline__ void test_and_stage(ST &st, const HitOut &H, const DevPwm &Pw, uint32_t motif, int64_t g,
                                               double fwd, double rev, int strand_mask, float floor32) {
    const double floor_ = (double) floor32;
    const bool try_f = (strand_mask & 1) && !(fwd < floor_);
    const bool try_r = (strand_mask & 2) && !(rev < floor_);
    if (!try_f && !try_r) return;
    const double2 mc = *reinterpret_cast<const double2 *>(Pw.thresh + 4 * (size_t) motif);      // {max_raw, cutoff}: one read
    const double max_raw = mc.x, cutoff = mc.y;
    if (try_f) {
        const double s = fwd / max_raw;
        if (s - cutoff >= -1e-10) stage_hit(st, H, motif, g, 0u, s);
    }
    if (try_r) {
        const double s = rev / max_raw;
        if (s - cutoff >= -1e-10) stage_hit(st, H, motif, g, 1u, s);
    }
}

// --------------------------------------------------------------------------- pack --

// One thread per 32 bases: two 16-byte loads, one 8-byte + one 4-byte store.
__global__ void __launch_bounds__(256) pack_kernel(const uint8_t *__restrict__ ascii, int64_t n_bases,
                                                   uint32_t *__restrict__ codes, uint32_t *__restrict__ nmask,
                                                   int64_t n_units, int aligned16) {
    const int64_t u = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_units) return;
    const int64_t base = u * 32;
    uint32_t raw[8];
    if (aligned16 && base + 32 <= n_bases) {
        const uint4 a = *reinterpret_cast<const uint4 *>(ascii + base);
        const uint4 b = *reinterpret_cast<const uint4 *>(ascii + base + 16);
        raw[0] = a.x; raw[1] = a.y; raw[2] = a.z; raw[3] = a.w;
        raw[4] = b.x; raw[5] = b.y; raw[6] = b.z; raw[7] = b.w;
    } else {
        for (int k = 0; k < 8; k++) {
            uint32_t w = 0;
            for (int j = 0; j < 4; j++) {
                const int64_t i = base + k * 4 + j;
                w |= (uint32_t) (i < n_bases ? ascii[i] : (uint8_t) 'A') << (8 * j);
            }
            raw[k] = w;
        }
    }
    uint64_t cw = 0;
    uint32_t nw = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t ch = ((raw[k] >> (8 * j)) & 0xFFu) | 0x20u;      // fold case (cscore.c:93-108)
            const uint32_t code = ((ch >> 1) ^ (ch >> 2)) & 3u;             // a,c,g,t -> 0,1,2,3
            const bool acgt = ch == 0x61u || ch == 0x63u || ch == 0x67u || ch == 0x74u;
            const int i = k * 4 + j;
            cw |= (uint64_t) (acgt ? code : 0u) << (2 * i);
            nw |= (acgt ? 0u : 1u) << i;
        }
    }
    if (base + 32 > n_bases) {          // bases past the end are neither N nor scanned
        const int valid = (int) (n_bases - base);
        nw &= low_mask(valid);
    }
    codes[2 * u] = (uint32_t) cw;
    codes[2 * u + 1] = (uint32_t) (cw >> 32);
    nmask[u] = nw;
}

// ---------------------------------------------------------------------- pre-filter --
//
// The pre-filter on the matrix cores (operand layout, quantisation and the proof that it never loses a hit: ms_internal.h,
// ms_plan.cpp).  Per wave and pass: 128 consecutive window starts (the double pass; 64 in the kernels with wide classes) = four (two)
// 32-column B operands per k-block -- the one-hot image of the lane's bases in fp4, fetched once per class from the pass's one-hot array and
// reused by every row tile -- and per row tile of 32 (motif, strand) rows ONE read of the A operand (fp6: 24 bytes per lane and k-block) and
// 2 NK matrix instructions per 64 window starts.  acc >= +0 (sign bit clear) in any of the 16 result registers of a lane marks a candidate
// (paired rows: bit 22 / bit 10 of the result, ms_internal.h).
//
// ---- candidate hand-off ----
// Candidates are ~2e-4 of the (window, motif) pairs.  One global atomic per find would put every wave of the chip on ONE address
// (measured in round 1: the whole kernel then runs at the ~90 M atomics/s a single word sustains).  Each wave therefore reserves
// BLOCKS of A.cand_block record slots of the global list (one atomicAdd per block) and stores its records straight into its block
// with ballot/mbcnt ranks; slots it leaves unused (fewer than 64 when a block is abandoned, the rest of the last block at the end)
// are written as empty records (flags 0), which rescore_kernel skips.  (Rounds 1-2 staged 64 records per wave in LDS and spilled
// them through a called function: at p = 1e-3, ten times the records, that flush was most of the kernel's time.)
// A record is per LANE and per table group: position, group, and one flag bit per field -- rescore_kernel expands the flags.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) i32x4 lds_i32x4;
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// What a wave keeps in registers about its candidates: the parking space (below) -- the rest of the hand-off lives in LDS (PfEmit)
struct MfWave {
    uint32_t rq;               // the wave's parking space: its LDS byte address (32-bit address arithmetic: a generic pointer made every entry's address a 64-bit multiply-add); rq_cap entries of kRareEntryWords words
    uint32_t rq_n;             // entries parked (wave-uniform)
    uint32_t rq_cap, rq_flush; // PfArgs::rare_cap; the fill at which the parked entries are decoded
};

// Where a class was when its wave's parking space ran full: row tile t is re-entered (its products are computed again) at operand
// `op` (0 / 1: the windows from g0 / from g0 + 32), past the first `skip` candidate lanes of that operand's event
struct PfResume {
    int t;
    uint32_t op, skip;
    uint32_t sub;              // double pass: the half (0: windows from pass0, 1: from pass0 + 64) whose event did not fit
};

// The wave's place in the global candidate list and the launch's constants, in LDS (one per wave: kPfEmitWords words): only the
// out-of-line decode (pf_flush) works with them, so none of it occupies registers of the scanning loop
struct PfEmit {
    unsigned long long base;   // next free slot of this wave's block in the global candidate list
    uint32_t left;             // slots left in the block
    uint32_t pad0;
    uint64_t *cand;
    unsigned long long *n_cand;
    uint64_t cand_cap, cand_static;
    uint32_t cand_block, pad[3];
};                             // (pf_flush reads it as seven 8-byte words)
static_assert(sizeof(PfEmit) == kPfEmitWords * sizeof(uint32_t), "PfEmit layout");

// bit n of the result = result register 15 - n is non-negative (field n of the lane's table group)
__device__ __forceinline__ uint32_t nonneg_flags(const f32x16 &c) {
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) m = __builtin_amdgcn_alignbit(m, (uint32_t) __float_as_int(c[j]), 31);   // (m << 1) | sign
    return ~m & 0xFFFFu;
}

// AND of the 16 sign bits (bit 31 of the result): clear <=> some result register is >= +0.  Eight 3-input ANDs (v_bitop3_b32)
// through one accumulator: the forwarded operand keeps the instruction at two fresh register reads
// (profiles/r03_insp_probe.log: 95 cycles per row tile against 108 for the tree of v_max3_i32 round 2 used).
__device__ __forceinline__ uint32_t all_negative(const f32x16 &c) {
    uint32_t x = (uint32_t) __float_as_int(c[0]) & (uint32_t) __float_as_int(c[1]) & (uint32_t) __float_as_int(c[2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2) x = x & (uint32_t) __float_as_int(c[i]) & (uint32_t) __float_as_int(c[i + 1]);
    return x & (uint32_t) __float_as_int(c[15]);
}

// One record per flagged lane into the wave's block of the global list (E: a register copy of the wave's PfEmit)
__device__ __forceinline__ void emit_rec(PfEmit &E, bool live, int64_t g, uint32_t flags, int32_t group) {
    const bool flagged = live && flags != 0;
    const unsigned long long mask = __ballot(flagged);
    if (mask == 0) return;
    const uint32_t n_new = (uint32_t) __popcll(mask);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
    if (n_new > E.left) {                                           // (cand_block >= 64 >= n_new: the next block always fits them)
        const uint32_t lane = threadIdx.x & 63u;
        if (lane < E.left && E.base + lane < E.cand_cap) E.cand[E.base + lane] = 0ULL;      // the abandoned rest of the block: empty records
        unsigned long long b = 0;
        if (lane == 0) b = E.cand_static + atomicAdd(E.n_cand, (unsigned long long) E.cand_block);
        E.base = ((unsigned long long) (uint32_t) __builtin_amdgcn_readfirstlane((int) (b >> 32)) << 32) |
                 (unsigned long long) (uint32_t) __builtin_amdgcn_readfirstlane((int) b);
        E.left = E.cand_block;
    }
    if (flagged && E.base + rank < E.cand_cap) E.cand[E.base + rank] = cand_pack((uint64_t) g, (uint32_t) group, flags);
    E.base += n_new;
    E.left -= n_new;
}

// bits of a paired row's 16 results: field X's flag is bit 22, field Y's bit 10 (ms_internal.h).  Two accumulators of eight registers
// each: m = 2 m | (c & mask) walks a register's two bits upwards one position per register (v_add_u32 + v_bitop3_b32, the fast
// VALU class: profiles/r03b_valu_rate.log).  fx / fy: bit n = result register 15 - n, as nonneg_flags.
__device__ __forceinline__ void pair_flags(const f32x16 &c, uint32_t &fx, uint32_t &fy) {
    uint32_t ma = 0, mb = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        ma = (ma + ma) | ((uint32_t) __float_as_int(c[j]) & kPairMask);
        mb = (mb + mb) | ((uint32_t) __float_as_int(c[8 + j]) & kPairMask);
    }
    // register j <= 7: X at bit 29 - j, Y at bit 17 - j of ma; register 8 + j: the same of mb
    fx = ((ma >> 14) & 0xFF00u) | ((mb >> 22) & 0xFFu);
    fy = ((ma >> 2) & 0xFF00u) | ((mb >> 10) & 0xFFu);
}

// ---- parked candidates ----
// A row tile holds a candidate in about one lane of its 64 windows x 32 rows, but decoding WHICH fields (one or two VALU operations
// per result register) and queueing the record costs the wave the same whether one lane needs it or all 64: in round 3's first
// paired kernel that was 28 % of the pre-filter (profiles/r03b_pf_pair.log), and -- inlined into every class -- the reason the
// kernel spilled registers it reloaded once per class and pass.  So the lanes that hold a candidate only PARK the flag bytes of their
// 16 result registers (park_store below; rounds 3-4: the registers themselves) and a two-word header (position, table group, kind) in
// the wave's LDS space -- two or three stores under the lanes' exec mask -- and when the space runs low the class returns to the one place that calls pf_flush, an ordinary
// (not inlined) function: it decodes the parked entries one per lane and queues the records.  An event with more candidate lanes
// than free entries parks what fits; the class comes back to the same row tile after the flush (PfResume).

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
// What a parked entry holds (round 5, second half): not the 16 result registers but the BYTES of them that carry the flags -- paired rows:
// bytes 1 and 2 of every register (bit 10 = field Y's flag, bit 22 = field X's): two registers per word, 8 words; plain rows: byte 3 (the sign),
// four registers per word, 4 words -- put together with v_perm_b32 (8 / 12 vector instructions per event and operand), then two (one) 16-byte
// stores and the 8-byte header instead of four and the header.  A parking store costs the wave ~55-65 cycles whatever its exec mask
// (profiles/r05d_pf_account.log: 2.6 k cycles per pass for 47 store instructions at p = 1e-4, 14.7 k for 220 at p = 1e-3), and a 48-byte
// entry instead of an 80-byte one gives the space 64 entries again beside the one-hot arrays.
__device__ __forceinline__ void park_pack(const f32x16 &c, uint32_t paired, u32x4 &p0, u32x4 &p1) {
    auto u = [&](int j) { return (uint32_t) __float_as_int(c[j]); };
    if (paired) {
        // word j = bytes {1, 2} of register j | bytes {1, 2} of register j + 8  (v_perm_b32: selector bytes 0-3 = the second source's, 4-7 = the first's)
#pragma unroll
        for (int j = 0; j < 4; j++) { p0[j] = __builtin_amdgcn_perm(u(j + 8), u(j), 0x06050201u); p1[j] = __builtin_amdgcn_perm(u(j + 12), u(j + 4), 0x06050201u); }
    } else {
        // word w = byte 3 of registers 4 w ... 4 w + 3
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const uint32_t lo = __builtin_amdgcn_perm(u(4 * w + 1), u(4 * w), 0x0C0C0703u), hi = __builtin_amdgcn_perm(u(4 * w + 3), u(4 * w + 2), 0x0C0C0703u);
            p0[w] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
        }
        p1 = u32x4{0u, 0u, 0u, 0u};
    }
}
__device__ __forceinline__ void park_put(uint32_t entry_lds, const u32x4 &p0, const u32x4 &p1, uint32_t paired, uint32_t hd0, uint32_t hd1) {
    lds_u32x4 *e = (lds_u32x4 *) (uintptr_t) entry_lds;
    e[0] = p0;
    if (paired) e[1] = p1;
    *(lds_u32x2 *) (e + 2) = u32x2{hd0, hd1};                                   // the header: always at byte 32
}
__device__ __forceinline__ void park_store(uint32_t entry_lds, const f32x16 &c, uint32_t paired, uint32_t hd0, uint32_t hd1) {
    u32x4 p0, p1;
    park_pack(c, paired, p0, p1);
    park_put(entry_lds, p0, p1, paired, hd0, hd1);
}

// Parks the event's candidate lanes number skip, skip + 1, ... while entries are free.  True: all parked (skip is 0 again).
__device__ __forceinline__ bool rare_park(MfWave &W, const f32x16 &c, bool hit, int64_t g, int32_t group, uint32_t paired, uint32_t &skip) {
    const unsigned long long mask = __ballot(hit);
    const uint32_t n_new = (uint32_t) __popcll(mask) - skip, n_free = W.rq_cap - W.rq_n;
    const uint32_t n_take = n_new < n_free ? n_new : n_free;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u)) - skip;   // (wraps for the lanes already parked)
    if (hit && rank < n_take)
        park_store(W.rq + __umul24(W.rq_n + rank, (uint32_t) (kRareEntryWords * 4)), c, paired, (uint32_t) g, (uint32_t) ((uint64_t) g >> 32) | ((uint32_t) group << 8) | (paired << 31));
    W.rq_n += n_take;
    if (n_take < n_new) { skip += n_take; return false; }
    skip = 0;
    return true;
}

// The candidate lanes of a row tile's two operands (hit0 / hit1) into the parking space; R says where to pick up after a flush.
// True: the space ran full, the class must leave for a flush and come back to this row tile.
__device__ __forceinline__ bool rare_park2(MfWave &W, PfResume &R, const f32x16 &c0, const f32x16 &c1, bool hit0, bool hit1, int64_t g0,
                                           int32_t group, uint32_t paired) {
    if (R.op == 0 && __any(hit0) && !rare_park(W, c0, hit0, g0, group, paired, R.skip)) return true;
    R.op = 1;
    if (__any(hit1) && !rare_park(W, c1, hit1, g0 + 32, group, paired, R.skip)) return true;
    R.op = 0;
    return false;
}

// The common case of an event, straight-line: BOTH operands' candidate lanes ranked at once (operand 0's lanes first, as rare_park2
// orders them) and parked with two exec-masked store sequences -- no resumable state, one branch.  Only an event that does not fit
// the free entries (or a class re-entered in the middle of one, R.op / R.skip set) takes rare_park2's piecewise form.  Round 4:
// profiles/r03zz_class_clock.log put the hand-off at ~5.7 k of a wave's ~24.5 k cycles per 64-window pass, ~600 cycles per event,
// most of it the dozen taken branches and scalar bookkeeping of the resumable form.
struct PfLive {                 // which of the lane's two windows lie inside the input: the lane's flags, and the wave's masks (scalar registers, made once per pass)
    bool l0, l1;
    unsigned long long m0, m1;
};
__device__ __forceinline__ bool park_both(MfWave &W, PfResume &R, const f32x16 &c0, const f32x16 &c1, const PfLive &L, bool cand0, bool cand1, int64_t g0,
                                          int32_t group, uint32_t paired, bool no_stores = false) {      // no_stores: measurement only (MS_PF_NOEMIT=5)
    // (the two conditions' masks ANDed as masks: a ballot of `live && cand` turns the AND into a register of 0 / 1 and compares that again --
    // four vector instructions per event, found in the ISA in round 5)
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(cand0) & L.m0, m1 = __builtin_amdgcn_ballot_w64(cand1) & L.m1;
    const bool hit0 = L.l0 && cand0, hit1 = L.l1 && cand1;
    const uint32_t n0 = (uint32_t) __popcll(m0), n1 = (uint32_t) __popcll(m1);
    if (__builtin_expect((R.op | R.skip) != 0u || W.rq_n + n0 + n1 > W.rq_cap, 0))
        return rare_park2(W, R, c0, c1, hit0, hit1, g0, group, paired);
    const uint32_t rank0 = __builtin_amdgcn_mbcnt_hi((uint32_t) (m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m0, 0u));
    const uint32_t rank1 = __builtin_amdgcn_mbcnt_hi((uint32_t) (m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m1, n0));
    const uint32_t hi = (uint32_t) ((uint64_t) g0 >> 32) | ((uint32_t) group << 8) | (paired << 31);     // (g0 + 32 never carries into bit 32: g0 < 2^34 is a multiple-of-64 base plus lane & 31)
    if (hit0 && !no_stores) park_store(W.rq + __umul24(W.rq_n + rank0, (uint32_t) (kRareEntryWords * 4)), c0, paired, (uint32_t) g0, hi);
    if (hit1 && !no_stores) park_store(W.rq + __umul24(W.rq_n + rank1, (uint32_t) (kRareEntryWords * 4)), c1, paired, (uint32_t) g0 + 32u, hi);
    W.rq_n += n0 + n1;
    return false;
}

// Decode and queue the n parked entries of a wave (lane i takes entry i).  All lanes of the wave, at a wave-uniform point.  NOT
// inlined, and everything it needs comes through two LDS addresses: its registers are its own business.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __attribute__((noinline)) void pf_flush(uint32_t em_lds, uint32_t rq_lds, uint32_t n) {
    const uint32_t lane = threadIdx.x & 63u;
    const bool mine = lane < n;
    lds_u32x2 *em2 = (lds_u32x2 *) (uintptr_t) em_lds;
    PfEmit E;
    {
        const u32x2 w0 = em2[0], w1 = em2[1], w2 = em2[2], w3 = em2[3], w4 = em2[4], w5 = em2[5], w6 = em2[6];
        E.base = ((unsigned long long) w0.y << 32) | w0.x;
        E.left = w1.x;
        E.cand = reinterpret_cast<uint64_t *>(((unsigned long long) w2.y << 32) | w2.x);
        E.n_cand = reinterpret_cast<unsigned long long *>(((unsigned long long) w3.y << 32) | w3.x);
        E.cand_cap = ((unsigned long long) w4.y << 32) | w4.x;
        E.cand_static = ((unsigned long long) w5.y << 32) | w5.x;
        E.cand_block = w6.x;
    }
    // the entry: 8 words of flag bytes (paired rows) or 4 (plain rows), then the header (park_store)
    u32x2 q[4] = {{0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}}, hd = {0u, 0u};
    if (mine) {
        lds_u32x2 *e = (lds_u32x2 *) (uintptr_t) (rq_lds + lane * (uint32_t) (kRareEntryWords * 4));
        q[0] = e[0]; q[1] = e[1]; q[2] = e[2]; q[3] = e[3];
        hd = e[4];
    }
    const bool paired = (hd.y >> 31) != 0u;
    const int64_t g = (int64_t) (((uint64_t) (hd.y & 0xFFu) << 32) | hd.x);
    const int32_t group = (int32_t) ((hd.y >> 8) & 0x3FFFu);
    // paired: word j = bits 8 ... 23 of result register j (low half) and of register j + 8 (high half): X's flag (bit 22) is bit 14 of its
    // half, Y's (bit 10) bit 2.  fx / fy: bit n = result register 15 - n.  Two accumulators walk the bits upwards one position per word.
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t w = (j & 1) ? q[j >> 1].y : q[j >> 1].x;
        lo = (lo + lo) | (w & 0x4004u);
        hi = (hi + hi) | ((w >> 16) & 0x4004u);
    }
    // register j <= 7: X at bit 21 - j, Y at bit 9 - j of lo; register 8 + j: the same of hi
    const uint32_t fx = ((lo >> 6) & 0xFF00u) | ((hi >> 14) & 0xFFu);
    const uint32_t fy = ((lo << 6) & 0xFF00u) | ((hi >> 2) & 0xFFu);
    // plain: word w = the top bytes of result registers 4 w ... 4 w + 3; bit n of the flags = register 15 - n is non-negative.  The four sign
    // bits of a word into a nibble (register 4 w first) by one multiplication: bits 0 / 8 / 16 / 24 x (2^27 + 2^18 + 2^9 + 1) meet at bits 27 ... 24
    uint32_t ms_ = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t t = (((w & 1) ? q[w >> 1].y : q[w >> 1].x) >> 7) & 0x01010101u;
        ms_ = (ms_ << 4) | ((t * 0x08040201u) >> 24 & 0xFu);
    }
    const uint32_t fs = ~ms_ & 0xFFFFu;
    emit_rec(E, mine, g, paired ? fx : fs, group);
    emit_rec(E, mine && paired, g, fy, group + 1);
    if (lane == 0) { em2[0] = u32x2{(uint32_t) E.base, (uint32_t) (E.base >> 32)}; ((lds_u32 *) (uintptr_t) em_lds)[2] = E.left; }
}

// OR of the 16 result patterns: some field of the lane is a candidate <=> (x & kPairMask) != 0.  v_bitop3_b32 by name: left alone
// hipcc picks v_or3_b32, which costs the SIMD 4.4 cycles against 2.7 (profiles/r03b_valu_rate.log)
__device__ __forceinline__ uint32_t or16(const f32x16 &c) {
    uint32_t x = __builtin_amdgcn_bitop3_b32((uint32_t) __float_as_int(c[0]), (uint32_t) __float_as_int(c[1]), (uint32_t) __float_as_int(c[2]), 0xFE);
#pragma unroll
    for (int i = 3; i < 15; i += 2) x = __builtin_amdgcn_bitop3_b32(x, (uint32_t) __float_as_int(c[i]), (uint32_t) __float_as_int(c[i + 1]), 0xFE);
    return x | (uint32_t) __float_as_int(c[15]);
}

// The lane's 8 bases of one k-block half as 32 fp4 one-hot k-slots: two reads of the 256-entry table (byte of four 2-bit codes ->
// 16 k-slots = 8 bytes).  n8 = the 8 bases' non-ACGT bits: such a base is an all-zero column (cscore.c:345-353 "adds nothing").
__device__ __forceinline__ i32x8 onehot_f4(const char *__restrict__ lut, uint32_t c16) {
    const int2 lo = *reinterpret_cast<const int2 *>(lut + ((c16 & 0xFFu) << 3)), hi = *reinterpret_cast<const int2 *>(lut + ((c16 >> 8) << 3));
    return i32x8{lo.x, lo.y, hi.x, hi.y, 0, 0, 0, 0};
}
__device__ __forceinline__ void clear_n(i32x8 &b, uint32_t n8) {
#pragma unroll
    for (int r = 0; r < 4; r++) {                                   // word r = bases 2r (low half) and 2r + 1 (high half)
        const uint32_t two = (n8 >> (2 * r)) & 3u;
        b[r] &= (int) (((two & 1u) ? 0u : 0xFFFFu) | ((two & 2u) ? 0u : 0xFFFF0000u));
    }
}

// What a pass keeps per lane for all its classes: the 64 bases (2-bit codes) from its two window starts g0 and g0 + 32.  Classes of 3
// or 4 k-blocks (motifs of 32 ... 63 columns: rare) read the 32 bases behind them themselves, and every class re-reads the
// non-ACGT bits where it needs them (rare): the narrow classes' hot loops own every register they can get.
struct PassSeq {
    uint64_t cw[2];
    const uint32_t *stg; // the wave's staged sequence words of this pass (LDS): 8 code words, then 4 non-ACGT words, from pass0 on
    bool any_n;          // wave-uniform: some lane sees a non-ACGT base in the 96 bases from g0
};

// ---- the double pass's one-hot array (the kernels for row tiles of 1 or 2 k-blocks; round 5) ----
// Every B operand of a pass is the fp4 one-hot image of EIGHT consecutive bases, 16 bytes, and the image of the bases from x on is the same
// whichever class, k-block, lane half or operand asks for it: paired half-block kb of the window at pass0 + r + 32 o + 64 s is entry
// r + 8 kb + 32 o + 64 s, plain k-block kb entry r + 16 kb + 8 h + 32 o + 64 s.  Rounds 2-4 (and the kernels with wide classes still) build each
// operand where it is used: two reads of the 256-entry table per operand (random 8-byte reads: the LDS array's only bank conflicts, 14 % of its
// busy cycles) and half a dozen vector instructions, ten operands per 64 windows on the benchmark plan.  A double pass builds entries 0 ... 159
// ONCE (lane l: entries l, 64 + l and 128 + (l & 31); non-ACGT bases cleared there), 2560 bytes per wave, and a class fetches its operands with
// one conflict-free ds_read_b128 each at a constant offset from the lane's entry.
constexpr int kOnehotEntries = 160;

// the 32 bases (2-bit codes) / their non-ACGT bits from window start pass0 + r + 32 i, cut out of the staged words (r = lane & 31):
// funnel shifts (v_alignbit_b32: a shift of 0 is the low operand itself, no special case)
__device__ __forceinline__ uint64_t staged_cw(const uint32_t *stg, uint32_t r, int i) {
    const uint32_t w = (r >> 4) + 2 * i, sh = (r & 15u) * 2u;
    const uint32_t lo = __builtin_amdgcn_alignbit(stg[w + 1], stg[w], sh), hi = __builtin_amdgcn_alignbit(stg[w + 2], stg[w + 1], sh);
    return ((uint64_t) hi << 32) | lo;
}
__device__ __forceinline__ uint32_t staged_nw(const uint32_t *stg, uint32_t r, int i) {
    return __builtin_amdgcn_alignbit(stg[9 + i], stg[8 + i], r);
}

// The matrix work of a row tile of TWO blocks, by name (round 4).  Such a tile lies in LDS lane-major, 48 bytes per lane (ms_internal.h,
// f6_word_off): three 16-byte reads land in TWELVE CONSECUTIVE registers, of which the first six are the block-0 operand and the last
// six the block-1 operand.  hipcc cannot place two 6-register operands on one run (its operands are separate 8-wide values) and moved
// four registers per row tile into place: 76 of a pass's ~940 vector instructions (profiles/r04_isa_account.md).  The reads are
// ds_read_b128 because the LDS array is this kernel's second-busiest unit (SQ_LDS_IDX_ACTIVE ~70 % of the CU's cycles): the same 48
// bytes per lane as three strided pair reads (ds_read2st64_b64 on a plane-major tile) take the array 24 cycles instead of 12 -- 16.5-16.6
// against 17.0 ms per 500 Mbase on one box, six ds_read_b64 16.8 (profiles/r04d_a_reads_ab.log).  Registers v[112:123] are this block's
// alone (clobbered); the waits are the compiler's own pattern -- `lgkmcnt(1)` before the first operand is needed (its six registers are
// the first two reads'), `lgkmcnt(0)` before the second's, and the 12 wait states (`s_nop 11`: hipcc's own pattern for this instruction in this binary) an 8-pass matrix instruction needs before a vector
// instruction reads its result.
// Round 5: the reads of the NEXT row tile's operand are issued inside the same block, right behind the last matrix instruction (which has
// read v[118:123] long before LDS data can come back), so that their latency -- a third of a wave's cycles were spent parked at
// s_waitcnt (SQ_WAIT_ANY, profiles/r05b_pmc_sq1.csv) -- runs beside the 12 wait states, the inspection and the loop branch instead of in
// front of the next products.  In C++ this was measured twice and lost (a second set of operand registers, moves); here the twelve
// registers are the same ones: the operand is dead once its instructions have issued.  The registers are an OPERAND of the blocks
// (`areg`, tied to v[112:123]), not a clobber: the compiler must keep them free between the blocks while the reads are in flight, and
// nothing but these blocks may touch them -- a_reads_begin starts the first row tile's reads, a_reads_drain waits for the last (unused)
// ones before the class returns and the registers go back to the compiler.  tests/test_host_cabi.py checks the built code for both.
// ---- the two builds of the pre-filter (round 6) ----
// DEFAULT (this macro undefined): NO hand-written blocks.  Row tiles of two blocks go through the same builtins the one-block tiles use (the
// A operand read per row tile with three ds_read_b128 by the compiler, its own waits, its own register allocation) and the hand-out uses the
// compiler's atomicAdd.  -DMS_PF_ASM builds the variant library (libmotifscan_amd_asm.so, MS_LIB_VARIANT=asm) WITH the blocks of rounds
// 4-5 above and below: they pin v[112:123] across compiler-made code and issue the hand-out's atomic behind the compiler's back, and what
// guards them is csrc/check_isa.py, a mandatory step of that variant's build that reads the code object back.
// Why the default flipped: measured side by side on two boxes in round 6 (profiles/r06a_bench_c4_noasm.json / _asm_same_box.json,
// profiles/r06b_e2e_ab_summary.log: kernel 15.72 / 15.72 ms, 15.34 / 15.34, 15.32 / 15.31; p = 1e-3 24.61 / 24.64) the two builds run the
// pre-filter in the same time -- the kernel is power-limited, and what the blocks saved in rounds 4-5 (register moves,
// exposed LDS latency) no longer shows -- while the asm form carries a hazard no test can rule out for a future compiler.  Same
// tables, same results bit for bit (tests/test_gpu_parity.py runs the goldens and configs[1] on both).  ms_build_flags() bit 0 = the blocks are in.
#ifdef MS_PF_ASM
constexpr bool kPfAsm = true;
#else
constexpr bool kPfAsm = false;
#endif
extern "C" int ms_build_flags(void) { return kPfAsm ? 1 : 0; }

// a two-block row tile's operands (lane-major, 48 bytes per lane: ms_internal.h) through ordinary LDS reads
__device__ __forceinline__ void load_a2(uint32_t pa, i32x8 &a0, i32x8 &a1) {
    const lds_i32x4 *q = (const lds_i32x4 *) (uintptr_t) pa;
    const i32x4 w0 = q[0], w1 = q[1], w2 = q[2];
    a0 = i32x8{w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], 0, 0};
    a1 = i32x8{w1[2], w1[3], w2[0], w2[1], w2[2], w2[3], 0, 0};
}
__device__ __forceinline__ i32x8 b_of(const i32x4 &b) { return i32x8{b[0], b[1], b[2], b[3], 0, 0, 0, 0}; }
__device__ __forceinline__ void pair_product2_intr(const i32x8 &a0, const i32x8 &a1, const i32x4 &b00, const i32x4 &b10, const i32x4 &b01, const i32x4 &b11,
                                                   int scale0, int scale1, const f32x16 &cc0, const f32x16 &cc1, f32x16 &c0, f32x16 &c1) {
    c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, b_of(b00), cc0, 2, 4, 0, scale0, 0, 127);
    c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, b_of(b10), cc1, 2, 4, 0, scale1, 0, 127);
    c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b_of(b01), c0, 2, 4, 0, scale0, 0, 127);
    c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b_of(b11), c1, 2, 4, 0, scale1, 0, 127);
}
__device__ __forceinline__ void plain_product2_intr(const i32x8 &a0, const i32x8 &a1, const i32x4 &b00, const i32x4 &b10, const i32x4 &b01, const i32x4 &b11,
                                                    f32x16 &c0, f32x16 &c1) {
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, b_of(b00), z, 2, 4, 0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, b_of(b10), z, 2, 4, 0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b_of(b01), c0, 2, 4, 0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b_of(b11), c1, 2, 4, 0, 0, 0, 0);
}

typedef int i32x12 __attribute__((ext_vector_type(12)));
__device__ __forceinline__ void a_reads_begin(uint32_t pa, i32x12 &areg) {
    asm volatile("ds_read_b128 v[112:115], %[pa]\n\t"
                 "ds_read_b128 v[116:119], %[pa] offset:16\n\t"
                 "ds_read_b128 v[120:123], %[pa] offset:32"
                 : "={v[112:123]}"(areg) : [pa] "v"(pa) : "memory");
}
__device__ __forceinline__ void a_reads_drain(i32x12 &areg) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "={v[112:123]}"(areg) : "0"(areg) : "memory");
}
// NEXT = byte distance to the next row tile's operand (the ds_read offset field: 16 bits)
// The single-pass form (the kernels WITH wide classes keep it): the row tile's reads, waits and products in one block, v[112:123] clobbered
// (the compiler may use them between the blocks: nothing is in flight there).
__device__ __forceinline__ void pair_product2_asm(uint32_t pa, const i32x4 &b00, const i32x4 &b10, const i32x4 &b01, const i32x4 &b11, int scale0, int scale1,
                                                  f32x16 &c0, f32x16 &c1) {
    const int one = 127;
    asm volatile("ds_read_b128 v[112:115], %[pa]\n\t"
                 "ds_read_b128 v[116:119], %[pa] offset:16\n\t"
                 "ds_read_b128 v[120:123], %[pa] offset:32\n\t"
                 "s_waitcnt lgkmcnt(1)\n\t"
                 "v_mfma_scale_f32_32x32x64_f8f6f4 %[c0], v[112:117], %[b00], 4.0, %[s0], %[s1] op_sel_hi:[0,0,0] cbsz:2 blgp:4\n\t"
                 "v_mfma_scale_f32_32x32x64_f8f6f4 %[c1], v[112:117], %[b10], 2.0, %[s0m], %[s1] op_sel_hi:[0,0,0] cbsz:2 blgp:4\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_mfma_scale_f32_32x32x64_f8f6f4 %[c0], v[118:123], %[b01], %[c0], %[s0], %[s1] op_sel_hi:[0,0,0] cbsz:2 blgp:4\n\t"
                 "v_mfma_scale_f32_32x32x64_f8f6f4 %[c1], v[118:123], %[b11], %[c1], %[s0m], %[s1] op_sel_hi:[0,0,0] cbsz:2 blgp:4\n\t"
                 "s_nop 11"
                 : [c0] "=&v"(c0), [c1] "=&v"(c1)
                 : [pa] "v"(pa), [b00] "v"(b00), [b10] "v"(b10), [b01] "v"(b01), [b11] "v"(b11), [s0] "v"(scale0), [s0m] "v"(scale1), [s1] "v"(one)
                 : "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "memory");
}
// ... and of plain rows: accumulators from 0, the instruction WITHOUT block scales (the scaled form is two instructions, v_mfma_ld_scale_b32 + the
// product, 16 bytes of code and two more register reads, for a scale of 2^0: -0.15 ms per 500 Mbase, profiles/r05_double_pass.log)
__device__ __forceinline__ void plain_product2_asm(uint32_t pa, const i32x4 &b00, const i32x4 &b10, const i32x4 &b01, const i32x4 &b11, f32x16 &c0, f32x16 &c1) {
    asm volatile("ds_read_b128 v[112:115], %[pa]\n\t"
                 "ds_read_b128 v[116:119], %[pa] offset:16\n\t"
                 "ds_read_b128 v[120:123], %[pa] offset:32\n\t"
                 "s_waitcnt lgkmcnt(1)\n\t"
                 "v_mfma_f32_32x32x64_f8f6f4 %[c0], v[112:117], %[b00], 0 cbsz:2 blgp:4\n\t"
                 "v_mfma_f32_32x32x64_f8f6f4 %[c1], v[112:117], %[b10], 0 cbsz:2 blgp:4\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_mfma_f32_32x32x64_f8f6f4 %[c0], v[118:123], %[b01], %[c0] cbsz:2 blgp:4\n\t"
                 "v_mfma_f32_32x32x64_f8f6f4 %[c1], v[118:123], %[b11], %[c1] cbsz:2 blgp:4\n\t"
                 "s_nop 11"
                 : [c0] "=&v"(c0), [c1] "=&v"(c1)
                 : [pa] "v"(pa), [b00] "v"(b00), [b10] "v"(b10), [b01] "v"(b01), [b11] "v"(b11)
                 : "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "memory");
}

// ---- the double pass (kernels without wide classes, round 5) ----
// A row tile's operand is read ONCE for 128 window starts: block a multiplies it with the B operands of the windows from pass0 and from
// pass0 + 32, the results are inspected, block b multiplies the SAME registers with those of pass0 + 64 and pass0 + 96 and then starts the
// next row tile's reads.  Why: the pre-filter is power-limited (tools/power_probe.py: ~1240 W of the board's 1400 W over a scan loop, the
// shader clock at 2.25 instead of 2.40 GHz), and with the operand reads of the two-block paired row tiles taken out the SAME cycle count ran
// at 2343 instead of 2188 MHz (profiles/r05_double_pass.log): the LDS reads cost clock, not cycles.  Half the reads, half the row-tile loop
// trips.  The accumulators are the same 32 registers for both halves.
__device__ __forceinline__ void pair_product2a_asm(i32x12 &areg, const i32x4 &b00, const i32x4 &b10, const i32x4 &b01, const i32x4 &b11, int scale0, int scale1,
                                                   f32x16 &c0, f32x16 &c1) {
    const int one = 127;
    asm volatile("s_waitcnt lgkmcnt(1)\n\t"
                 "v_mfma_scale_f32_32x32x64_f8f6f4 %[c0], v[112:117], %[b00], 4.0, %[s0], %[s1] op_sel_hi:[0,0,0] cbsz:2 blgp:4\n\t"
                 "v_mfma_scale_f32_32x32x64_f8f6f4 %[c1], v[112:117], %[b10], 2.0, %[s0m], %[s1] op_sel_hi:[0,0,0] cbsz:2 blgp:4\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_mfma_scale_f32_32x32x64_f8f6f4 %[c0], v[118:123], %[b01], %[c0], %[s0], %[s1] op_sel_hi:[0,0,0] cbsz:2 blgp:4\n\t"
                 "v_mfma_scale_f32_32x32x64_f8f6f4 %[c1], v[118:123], %[b11], %[c1], %[s0m], %[s1] op_sel_hi:[0,0,0] cbsz:2 blgp:4\n\t"
                 "s_nop 11"
                 : [c0] "=&v"(c0), [c1] "=&v"(c1), "={v[112:123]}"(areg)
                 : [b00] "v"(b00), [b10] "v"(b10), [b01] "v"(b01), [b11] "v"(b11), [s0] "v"(scale0), [s0m] "v"(scale1), [s1] "v"(one), "2"(areg)
                 : "memory");
}
template <int NEXT>
__device__ __forceinline__ void pair_product2b_asm(uint32_t pa, i32x12 &areg, const i32x4 &b00, const i32x4 &b10, const i32x4 &b01, const i32x4 &b11, int scale0, int scale1,
                                                   f32x16 &c0, f32x16 &c1) {
    const int one = 127;
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %[c0], v[112:117], %[b00], 4.0, %[s0], %[s1] op_sel_hi:[0,0,0] cbsz:2 blgp:4\n\t"
                 "v_mfma_scale_f32_32x32x64_f8f6f4 %[c1], v[112:117], %[b10], 2.0, %[s0m], %[s1] op_sel_hi:[0,0,0] cbsz:2 blgp:4\n\t"
                 "v_mfma_scale_f32_32x32x64_f8f6f4 %[c0], v[118:123], %[b01], %[c0], %[s0], %[s1] op_sel_hi:[0,0,0] cbsz:2 blgp:4\n\t"
                 "v_mfma_scale_f32_32x32x64_f8f6f4 %[c1], v[118:123], %[b11], %[c1], %[s0m], %[s1] op_sel_hi:[0,0,0] cbsz:2 blgp:4\n\t"
                 "ds_read_b128 v[112:115], %[pa] offset:%[n0]\n\t"
                 "ds_read_b128 v[116:119], %[pa] offset:%[n1]\n\t"
                 "ds_read_b128 v[120:123], %[pa] offset:%[n2]\n\t"
                 "s_nop 11"
                 : [c0] "=&v"(c0), [c1] "=&v"(c1), "={v[112:123]}"(areg)
                 : [pa] "v"(pa), [b00] "v"(b00), [b10] "v"(b10), [b01] "v"(b01), [b11] "v"(b11), [s0] "v"(scale0), [s0m] "v"(scale1), [s1] "v"(one), "2"(areg),
                   [n0] "n"(NEXT), [n1] "n"(NEXT + 16), [n2] "n"(NEXT + 32)
                 : "memory");
}
__device__ __forceinline__ void plain_product2a_asm(i32x12 &areg, const i32x4 &b00, const i32x4 &b10, const i32x4 &b01, const i32x4 &b11, f32x16 &c0, f32x16 &c1) {
    asm volatile("s_waitcnt lgkmcnt(1)\n\t"
                 "v_mfma_f32_32x32x64_f8f6f4 %[c0], v[112:117], %[b00], 0 cbsz:2 blgp:4\n\t"
                 "v_mfma_f32_32x32x64_f8f6f4 %[c1], v[112:117], %[b10], 0 cbsz:2 blgp:4\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_mfma_f32_32x32x64_f8f6f4 %[c0], v[118:123], %[b01], %[c0] cbsz:2 blgp:4\n\t"
                 "v_mfma_f32_32x32x64_f8f6f4 %[c1], v[118:123], %[b11], %[c1] cbsz:2 blgp:4\n\t"
                 "s_nop 11"
                 : [c0] "=&v"(c0), [c1] "=&v"(c1), "={v[112:123]}"(areg)
                 : [b00] "v"(b00), [b10] "v"(b10), [b01] "v"(b01), [b11] "v"(b11), "2"(areg)
                 : "memory");
}
template <int NEXT>
__device__ __forceinline__ void plain_product2b_asm(uint32_t pa, i32x12 &areg, const i32x4 &b00, const i32x4 &b10, const i32x4 &b01, const i32x4 &b11, f32x16 &c0, f32x16 &c1) {
    asm volatile("v_mfma_f32_32x32x64_f8f6f4 %[c0], v[112:117], %[b00], 0 cbsz:2 blgp:4\n\t"
                 "v_mfma_f32_32x32x64_f8f6f4 %[c1], v[112:117], %[b10], 0 cbsz:2 blgp:4\n\t"
                 "v_mfma_f32_32x32x64_f8f6f4 %[c0], v[118:123], %[b01], %[c0] cbsz:2 blgp:4\n\t"
                 "v_mfma_f32_32x32x64_f8f6f4 %[c1], v[118:123], %[b11], %[c1] cbsz:2 blgp:4\n\t"
                 "ds_read_b128 v[112:115], %[pa] offset:%[n0]\n\t"
                 "ds_read_b128 v[116:119], %[pa] offset:%[n1]\n\t"
                 "ds_read_b128 v[120:123], %[pa] offset:%[n2]\n\t"
                 "s_nop 11"
                 : [c0] "=&v"(c0), [c1] "=&v"(c1), "={v[112:123]}"(areg)
                 : [pa] "v"(pa), [b00] "v"(b00), [b10] "v"(b10), [b01] "v"(b01), [b11] "v"(b11), "2"(areg),
                   [n0] "n"(NEXT), [n1] "n"(NEXT + 16), [n2] "n"(NEXT + 32)
                 : "memory");
}

// All row tiles of one class of plain rows (NK k-blocks each).
template <int NK, bool MEAS>
__device__ __forceinline__ void f6_class(const PfArgs &A, MfWave &W, const char *__restrict__ lds, const char *__restrict__ lut,
                                         uint32_t byte_off, int n_row_tiles, int32_t first_group, const PassSeq &Q,
                                         int64_t pass0, const PfLive &L, PfResume &R) {
    // pass0 = the pass's first window start (wave-uniform: scalar registers); this lane's two window starts are pass0 + r and
    // pass0 + r + 32 with r = lane & 31 -- recomputed where needed (rare paths), not carried
    const uint32_t lane = threadIdx.x & 63u, h = lane >> 5;
    // R.t: the row tile to start at; the class returns early, with the row tile to come back to in R, when the wave's parking space
    // runs low or full (the caller has the parked entries decoded -- ONE call site for all classes -- and comes back)
    constexpr int kStep = NK * kF6BytesPerKb;
    const char *p = lds + byte_off + lane * (NK == 2 ? 48u : 8u) + (uint32_t) R.t * (uint32_t) kStep;
    constexpr int NW = NK > 2 ? 3 : 2;
    // B operands: k-block kb of the window at g0 covers bases 16 kb + 8 h ... + 7 from g0; of the window at g0 + 32 the same from there
    i32x8 b0[NK], b1[NK];
    uint64_t cw[NW];
    cw[0] = Q.cw[0];
    cw[1] = Q.cw[1];
    if constexpr (NW == 3) cw[2] = staged_cw(Q.stg, lane & 31u, 2);
#pragma unroll
    for (int kb = 0; kb < NK; kb++) {
        const int w = kb >> 1, sh = 32 * (kb & 1);
        b0[kb] = onehot_f4(lut, (uint32_t) (cw[w] >> (sh + 16 * h)) & 0xFFFFu);
        b1[kb] = onehot_f4(lut, (uint32_t) (cw[w + 1] >> (sh + 16 * h)) & 0xFFFFu);
    }
    if (Q.any_n) {                                                                // rare, wave-uniform
        uint32_t nw[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) nw[i] = staged_nw(Q.stg, lane & 31u, i);
#pragma unroll
        for (int kb = 0; kb < NK; kb++) {
            const int w = kb >> 1, sh = 16 * (kb & 1);
            const uint32_t keep = (kb == NK - 1 && h) ? 0x7Fu : 0xFFu;            // the row tile's last column carries the bias: never cleared
            clear_n(b0[kb], (nw[w] >> (sh + 8 * h)) & keep);
            clear_n(b1[kb], (nw[w + 1] >> (sh + 8 * h)) & keep);
        }
    }
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto product = [&](const char *q, f32x16 &c0, f32x16 &c1) {
        i32x8 a[NK];
#pragma unroll
        for (int kb = 0; kb < NK; kb++) {
            const int2 w0 = *reinterpret_cast<const int2 *>(q + kb * kF6BytesPerKb);
            const int2 w1 = *reinterpret_cast<const int2 *>(q + kb * kF6BytesPerKb + 512);
            const int2 w2 = *reinterpret_cast<const int2 *>(q + kb * kF6BytesPerKb + 1024);
            a[kb] = i32x8{w0.x, w0.y, w1.x, w1.y, w2.x, w2.y, 0, 0};
        }
        // (scale operands 0, 0: the compiler emits the instruction without block scales, v_mfma_f32_32x32x64_f8f6f4 -- scale 2^0)
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], b0[0], z, 2, 4, 0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], b1[0], z, 2, 4, 0, 0, 0, 0);
#pragma unroll
        for (int kb = 1; kb < NK; kb++) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[kb], b0[kb], c0, 2, 4, 0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[kb], b1[kb], c1, 2, 4, 0, 0, 0, 0);
        }
    };
    // one row tile in flight per wave: with paired rows most instructions belong to four-instruction row tiles, and a second set of
    // 32 accumulators (round 2's two tiles in flight for the narrow classes) costs the whole kernel its registers
    // (ONE loop exit: a class that must leave early -- the parking space ran full inside row tile t: come back to it; or runs low:
    // come back to t + 1 -- sets `back` and ends the loop through its counter, so the hot path is product, inspection, one branch)
    int n_run = n_row_tiles;
    if constexpr (MEAS) { if (A.no_emit == 2) n_run = 0; }                          // measurement: the per-pass and per-class set-up alone
    int back = n_row_tiles;
    [[maybe_unused]] uint32_t pa = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) const char *) p;       // the row tile's LDS address
    [[maybe_unused]] i32x4 bq[4];
    if constexpr (NK == 2) {
        bq[0] = i32x4{b0[0][0], b0[0][1], b0[0][2], b0[0][3]}; bq[1] = i32x4{b1[0][0], b1[0][1], b1[0][2], b1[0][3]};
        bq[2] = i32x4{b0[1][0], b0[1][1], b0[1][2], b0[1][3]}; bq[3] = i32x4{b1[1][0], b1[1][1], b1[1][2], b1[1][3]};
    }
    for (int t = R.t; t < n_run; t++, p += kStep, pa += (uint32_t) kStep) {
        f32x16 c0, c1;
        if constexpr (NK == 2 && kPfAsm) plain_product2_asm(pa, bq[0], bq[1], bq[2], bq[3], c0, c1);
        else if constexpr (NK == 2) { i32x8 a0, a1; load_a2(pa, a0, a1); plain_product2_intr(a0, a1, bq[0], bq[1], bq[2], bq[3], c0, c1); }
        else product(p, c0, c1);
        if constexpr (MEAS) { if (A.no_emit == 3) { asm volatile("" : : "v"(c0), "v"(c1)); continue; } }      // measurement: operand reads + products, no inspection
        const uint32_t x0 = all_negative(c0), x1 = all_negative(c1);
        if (__builtin_expect(__any((int) (x0 & x1) >= 0) && !(MEAS && A.no_emit >= 1 && A.no_emit <= 3), 0)) {
            // rare path (about one row tile in four holds a candidate in some lane): the candidate lanes park their results
            const bool full = park_both(W, R, c0, c1, L, (int) x0 >= 0, (int) x1 >= 0, pass0 + (lane & 31u), first_group + 2 * t + (int32_t) h, 0u,
                                        MEAS && A.no_emit == 5);
            if constexpr (MEAS) { if (A.no_emit >= 4) W.rq_n = 0; }              // measurement: the events run, their entries are dropped (4), nor stored at all (5): no decode
            if (full || W.rq_n >= W.rq_flush) { back = full ? t : t + 1; t = n_run; }
        }
    }
    R.t = back;
}

// All row tiles of one class of PAIRED rows (ms_internal.h): NK half-blocks of 8 columns, k-half 0 = field X, k-half 1 = field Y (block
// scales 2^-6 / 2^-18), both k-halves of the B operand = the same 8 bases, accumulators started at the inline constant 4.0, the bias
// column's B slots constant.  A row tile answers for 32 motifs x 2 strands with the 32 result registers that answer for 16 in a
// plain row tile.
template <int NK, bool MEAS>
__device__ __forceinline__ void f6_pair_class(const PfArgs &A, MfWave &W, const char *__restrict__ lds, const char *__restrict__ lut,
                                              uint32_t byte_off, int n_row_tiles, int32_t first_group, const PassSeq &Q,
                                              int64_t pass0, const PfLive &L, PfResume &R) {
    const uint32_t lane = threadIdx.x & 63u, h = lane >> 5;
    constexpr int kStep = NK * kF6BytesPerKb;
    const char *p = lds + byte_off + lane * (NK == 2 ? 48u : 8u) + (uint32_t) R.t * (uint32_t) kStep;      // R: see f6_class
    // B operands: half-block kb covers bases 8 kb ... 8 kb + 7 of the window, in both lane halves
    i32x8 b0[NK], b1[NK];
#pragma unroll
    for (int kb = 0; kb < NK; kb++) {
        b0[kb] = onehot_f4(lut, (uint32_t) (Q.cw[0] >> (16 * kb)) & 0xFFFFu);
        b1[kb] = onehot_f4(lut, (uint32_t) (Q.cw[1] >> (16 * kb)) & 0xFFFFu);
    }
    if (Q.any_n) {                                                                // rare, wave-uniform
        const uint32_t nw0 = staged_nw(Q.stg, lane & 31u, 0), nw1 = staged_nw(Q.stg, lane & 31u, 1);
#pragma unroll
        for (int kb = 0; kb < NK; kb++) {
            const uint32_t keep = kb == NK - 1 ? 0x7Fu : 0xFFu;                   // the fields' last column carries the bias: never cleared
            clear_n(b0[kb], (nw0 >> (8 * kb)) & keep);
            clear_n(b1[kb], (nw1 >> (8 * kb)) & keep);
        }
    }
    // the bias column (last column of the last half-block): constant k-slots in place of the base's one-hot image
    b0[NK - 1][3] = (int) (((uint32_t) b0[NK - 1][3] & 0xFFFFu) | (kPairBiasB << 16));
    b1[NK - 1][3] = (int) (((uint32_t) b1[NK - 1][3] & 0xFFFFu) | (kPairBiasB << 16));
    // The two products of a row tile start from DIFFERENT inline constants, 4.0 and 2.0, with block scales one binade apart: the same
    // mantissa layout either way (a constant shared by two instructions is put into 16 registers by hipcc, eight v_mov per row tile)
    const int scale0 = h ? kPairScaleY : kPairScaleX, scale1 = scale0 - 1;
    f32x16 cc0, cc1;
#pragma unroll
    for (int j = 0; j < 16; j++) { cc0[j] = kPairC; cc1[j] = 0.5f * kPairC; }
    auto load_a = [&](const char *q, int kb) {
        const int2 w0 = *reinterpret_cast<const int2 *>(q + kb * kF6BytesPerKb);
        const int2 w1 = *reinterpret_cast<const int2 *>(q + kb * kF6BytesPerKb + 512);
        const int2 w2 = *reinterpret_cast<const int2 *>(q + kb * kF6BytesPerKb + 1024);
        return i32x8{w0.x, w0.y, w1.x, w1.y, w2.x, w2.y, 0, 0};
    };
    auto product = [&](const char *q, f32x16 &c0, f32x16 &c1) {
        // (three half-blocks: the A operands are read one or two at a time -- 6 B operands of 4 registers, 3 A operands of 6 and the
        // 32 results do not fit beside the rest)
        i32x8 a = load_a(q, 0);
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b0[0], cc0, 2, 4, 0, scale0, 0, 127);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b1[0], cc1, 2, 4, 0, scale1, 0, 127);
#pragma unroll
        for (int kb = 1; kb < NK; kb++) {
            a = load_a(q, kb);
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b0[kb], c0, 2, 4, 0, scale0, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b1[kb], c1, 2, 4, 0, scale1, 0, 127);
        }
    };
    int n_run = n_row_tiles;
    if constexpr (MEAS) { if (A.no_emit == 2) n_run = 0; }
    int back = n_row_tiles;                                                         // (one loop exit: see f6_class)
    [[maybe_unused]] uint32_t pa = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) const char *) p;       // the row tile's LDS address
    [[maybe_unused]] i32x4 bq[4];
    if constexpr (NK == 2) {
        bq[0] = i32x4{b0[0][0], b0[0][1], b0[0][2], b0[0][3]}; bq[1] = i32x4{b1[0][0], b1[0][1], b1[0][2], b1[0][3]};
        bq[2] = i32x4{b0[1][0], b0[1][1], b0[1][2], b0[1][3]}; bq[3] = i32x4{b1[1][0], b1[1][1], b1[1][2], b1[1][3]};
    }
    for (int t = R.t; t < n_run; t++, p += kStep, pa += (uint32_t) kStep) {
        f32x16 c0, c1;
        if constexpr (NK == 2 && kPfAsm) pair_product2_asm(pa, bq[0], bq[1], bq[2], bq[3], scale0, scale1, c0, c1);
        else if constexpr (NK == 2) { i32x8 a0, a1; load_a2(pa, a0, a1); pair_product2_intr(a0, a1, bq[0], bq[1], bq[2], bq[3], scale0, scale1, cc0, cc1, c0, c1); }
        else product(p, c0, c1);
        if constexpr (MEAS) { if (A.no_emit == 3) { asm volatile("" : : "v"(c0), "v"(c1)); continue; } }
        const uint32_t x0 = or16(c0), x1 = or16(c1);
        if (__builtin_expect(__any(((x0 | x1) & kPairMask) != 0u) && !(MEAS && A.no_emit >= 1 && A.no_emit <= 3), 0)) {
            // rare path: the candidate lanes park their results (table groups 4 t + 2 h for field X and + 1 for field Y)
            const bool full = park_both(W, R, c0, c1, L, (x0 & kPairMask) != 0u, (x1 & kPairMask) != 0u, pass0 + (lane & 31u),
                                        first_group + 4 * t + 2 * (int32_t) h, 1u, MEAS && A.no_emit == 5);
            if constexpr (MEAS) { if (A.no_emit >= 4) W.rq_n = 0; }
            if (full || W.rq_n >= W.rq_flush) { back = full ? t : t + 1; t = n_run; }
        }
    }
    R.t = back;
}

// ---- dense candidates (round 5): the kernel for cutoffs at which nearly every row tile holds candidates ----
// Parking pays while an event is a lane or a few in a row tile (p = 1e-4: 0.4 hits per 64 windows x 32 rows; p = 1e-3: 4.4 -- there the parked
// form still wins, 31.9 against 34.9 ms per 500 Mbase).  At p = 1e-2 a row tile holds 44: a third of the lanes of both operands park, the
// space is decoded after every event and the hand-off is most of the kernel.  There the flags are decoded IN PLACE for the whole wave -- the
// vector work that costs the same for one lane or 64 is well used -- and the records go straight into the wave's block of the list, its
// place kept in scalar registers (no parking space, no pf_flush): 5.7 against 9.5 ms on the 62.5-Mbase shard (profiles/r05_dense_form.log).
// scan_locked launches this instantiation when the PREVIOUS scan of the PWM set at these cutoffs and strands found more than
// kDenseHitsPerHalfTile hits per row tile and 64 windows.
struct PfOut {
    unsigned long long base;   // next free slot of this wave's block in the global candidate list
    uint32_t left;             // slots left in the block
};
// one record per flagged lane into the wave's block (ranks by ballot / mbcnt; a block that cannot take them all is abandoned: its rest becomes
// empty records, the next one comes from the counter)
__device__ __forceinline__ void put_recs(const PfArgs &A, PfOut &O, bool flagged, uint64_t rec) {
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(flagged);
    if (mask == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_new = (uint32_t) __popcll(mask);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
    if (n_new > O.left) {                                           // (cand_block >= 64 >= n_new: the next block always fits them)
        if (lane < O.left && O.base + lane < A.cand_cap) A.cand[O.base + lane] = 0ULL;
        unsigned long long b = 0;
        if (lane == 0) b = A.cand_static + atomicAdd(A.n_cand, (unsigned long long) A.cand_block);
        O.base = ((unsigned long long) (uint32_t) __builtin_amdgcn_readfirstlane((int) (b >> 32)) << 32) |
                 (unsigned long long) (uint32_t) __builtin_amdgcn_readfirstlane((int) b);
        O.left = A.cand_block;
    }
    if (flagged && O.base + rank < A.cand_cap) A.cand[O.base + rank] = rec;
    O.base += n_new;
    O.left -= n_new;
}
template <bool PAIRED>
__device__ __forceinline__ void dense_event(const PfArgs &A, PfOut &O, const f32x16 &c0, const f32x16 &c1, const PfLive &L, int64_t g0, int32_t group) {
    if constexpr (PAIRED) {
        uint32_t x0, y0, x1, y1;
        pair_flags(c0, x0, y0);
        pair_flags(c1, x1, y1);
        put_recs(A, O, L.l0 && x0 != 0u, cand_pack((uint64_t) g0, (uint32_t) group, x0));
        put_recs(A, O, L.l0 && y0 != 0u, cand_pack((uint64_t) g0, (uint32_t) group + 1u, y0));
        put_recs(A, O, L.l1 && x1 != 0u, cand_pack((uint64_t) g0 + 32u, (uint32_t) group, x1));
        put_recs(A, O, L.l1 && y1 != 0u, cand_pack((uint64_t) g0 + 32u, (uint32_t) group + 1u, y1));
    } else {
        const uint32_t f0 = nonneg_flags(c0), f1 = nonneg_flags(c1);
        put_recs(A, O, L.l0 && f0 != 0u, cand_pack((uint64_t) g0, (uint32_t) group, f0));
        put_recs(A, O, L.l1 && f1 != 0u, cand_pack((uint64_t) g0 + 32u, (uint32_t) group, f1));
    }
}

// ---- classes of a double pass ----
struct PfLive2 { PfLive h[2]; };       // the halves of a double pass: windows from pass0 / from pass0 + 64

// What the halves of a row tile share: inspection result -> the candidate lanes park.  Returns true when the parking space ran full
// inside this half (the class leaves and comes back to row tile t, half s); sets `low` when the space runs low (the class leaves after t).
template <bool PAIRED, bool MEAS>
__device__ __forceinline__ bool half_event(const PfArgs &A, MfWave &W, PfResume &R, const f32x16 &c0, const f32x16 &c1, const PfLive &L, uint32_t x0, uint32_t x1,
                                           int64_t g0, int32_t group, bool &low) {
    const bool cand0 = PAIRED ? (x0 & kPairMask) != 0u : (int) x0 >= 0, cand1 = PAIRED ? (x1 & kPairMask) != 0u : (int) x1 >= 0;
    const bool full = park_both(W, R, c0, c1, L, cand0, cand1, g0, group, PAIRED ? 1u : 0u, MEAS && A.no_emit == 5);
    if constexpr (MEAS) { if (A.no_emit >= 4) W.rq_n = 0; }                       // measurement: the events run, their entries are dropped (4), nor stored at all (5): no decode
    low = low || W.rq_n >= W.rq_flush;
    return full;
}

// All row tiles of one class of PAIRED rows against the 128 window starts of a double pass (hw: the wave's one-hot array of the pass).
template <int NK, bool MEAS, bool DENSE, int FLOOR = 0>
__device__ __forceinline__ void f6_pair_class2(const PfArgs &A, MfWave &W, PfOut &O, const char *__restrict__ lds, uint32_t byte_off, int n_row_tiles, int32_t first_group,
                                               uint32_t hw, int64_t pass0, const PfLive2 &L, PfResume &R) {
    static_assert(NK == 1 || NK == 2, "paired rows have one or two half-blocks");
    const uint32_t lane = threadIdx.x & 63u, h = lane >> 5, r = lane & 31u;
    constexpr int kStep = NK * kF6BytesPerKb;
    const char *p = lds + byte_off + lane * (NK == 2 ? 48u : 8u) + (uint32_t) R.t * (uint32_t) kStep;
    // B operands: half-block kb of the window at pass0 + r + 32 o + 64 s = entry r + 8 kb + 32 o + 64 s of the one-hot array
    i32x4 b[2][2][NK];
    {
        const lds_i32x4 *e = (const lds_i32x4 *) (uintptr_t) (hw + r * 16u);
#pragma unroll
        for (int sg = 0; sg < 2; sg++)
#pragma unroll
            for (int o = 0; o < 2; o++) {
#pragma unroll
                for (int kb = 0; kb < NK; kb++) b[sg][o][kb] = e[8 * kb + 32 * o + 64 * sg];
                // the bias column (last column of the last half-block): constant k-slots in place of the base's one-hot image
                b[sg][o][NK - 1][3] = (int) (((uint32_t) b[sg][o][NK - 1][3] & 0xFFFFu) | (kPairBiasB << 16));
            }
    }
    const int scale0 = h ? kPairScaleY : kPairScaleX, scale1 = scale0 - 1;         // (see f6_pair_class)
    f32x16 cc0, cc1;
#pragma unroll
    for (int j = 0; j < 16; j++) { cc0[j] = kPairC; cc1[j] = 0.5f * kPairC; }
    int n_run = n_row_tiles;
    if constexpr (MEAS) { if (A.no_emit == 2) n_run = 0; }
    if constexpr (FLOOR == 4) n_run = 0;                                             // floor instantiation: the per-pass and per-class set-up alone
    int back = n_row_tiles;
    [[maybe_unused]] uint32_t pa = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) const char *) p;
    [[maybe_unused]] i32x12 areg;
    if constexpr (NK == 2 && kPfAsm) a_reads_begin(pa, areg);
    const bool skip_events = MEAS && A.no_emit >= 1 && A.no_emit <= 3;
    [[maybe_unused]] i32x8 a_fix, a1_fix;                                            // FLOOR 3: ONE operand read per class, the matrix instructions alone in the loop
    if constexpr (FLOOR == 3) { if constexpr (NK == 2) load_a2(pa, a_fix, a1_fix); else { const int2 w0 = *reinterpret_cast<const int2 *>(p), w1 = *reinterpret_cast<const int2 *>(p + 512), w2 = *reinterpret_cast<const int2 *>(p + 1024); a_fix = i32x8{w0.x, w0.y, w1.x, w1.y, w2.x, w2.y, 0, 0}; a1_fix = a_fix; } }
    for (int t = R.t; t < n_run; t++, p += kStep, pa += (uint32_t) kStep) {
        f32x16 c0, c1;
        [[maybe_unused]] i32x8 a, a1;
        bool stop = false, low = false;
        if constexpr (NK == 2 && kPfAsm) pair_product2a_asm(areg, b[0][0][0], b[0][1][0], b[0][0][1], b[0][1][1], scale0, scale1, c0, c1);
        else if constexpr (NK == 2) { if constexpr (FLOOR == 3) { a = a_fix; a1 = a1_fix; } else load_a2(pa, a, a1); pair_product2_intr(a, a1, b[0][0][0], b[0][1][0], b[0][0][1], b[0][1][1], scale0, scale1, cc0, cc1, c0, c1); }
        else {
            if constexpr (FLOOR == 3) a = a_fix;
            else {
                const int2 w0 = *reinterpret_cast<const int2 *>(p), w1 = *reinterpret_cast<const int2 *>(p + 512), w2 = *reinterpret_cast<const int2 *>(p + 1024);
                a = i32x8{w0.x, w0.y, w1.x, w1.y, w2.x, w2.y, 0, 0};
            }
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, i32x8{b[0][0][0][0], b[0][0][0][1], b[0][0][0][2], b[0][0][0][3], 0, 0, 0, 0}, cc0, 2, 4, 0, scale0, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, i32x8{b[0][1][0][0], b[0][1][0][1], b[0][1][0][2], b[0][1][0][3], 0, 0, 0, 0}, cc1, 2, 4, 0, scale1, 0, 127);
        }
        if constexpr (MEAS) { if (A.no_emit == 3) asm volatile("" : : "v"(c0), "v"(c1)); }
        if constexpr (FLOOR >= 2) asm volatile("" : : "v"(c0), "v"(c1));          // floor instantiations (below): no inspection
        else if constexpr (FLOOR == 1) { const uint32_t x0 = or16(c0), x1 = or16(c1); asm volatile("" : : "s"(__builtin_amdgcn_ballot_w64(((x0 | x1) & kPairMask) != 0u))); }   // inspection, no hand-off
        else if (R.sub == 0u && !(MEAS && A.no_emit == 3)) {
            const uint32_t x0 = or16(c0), x1 = or16(c1);
            if (__builtin_expect(__any(((x0 | x1) & kPairMask) != 0u) && !skip_events, DENSE ? 1 : 0)) {
                if constexpr (DENSE) dense_event<true>(A, O, c0, c1, L.h[0], pass0 + r, first_group + 4 * t + 2 * (int32_t) h);
                else if (half_event<true, MEAS>(A, W, R, c0, c1, L.h[0], x0, x1, pass0 + r, first_group + 4 * t + 2 * (int32_t) h, low)) { back = t; R.sub = 0u; stop = true; }
            }
        }
        if (!stop) {
            if constexpr (NK == 2 && kPfAsm) pair_product2b_asm<kStep>(pa, areg, b[1][0][0], b[1][1][0], b[1][0][1], b[1][1][1], scale0, scale1, c0, c1);
            else if constexpr (NK == 2) pair_product2_intr(a, a1, b[1][0][0], b[1][1][0], b[1][0][1], b[1][1][1], scale0, scale1, cc0, cc1, c0, c1);
            else {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, i32x8{b[1][0][0][0], b[1][0][0][1], b[1][0][0][2], b[1][0][0][3], 0, 0, 0, 0}, cc0, 2, 4, 0, scale0, 0, 127);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, i32x8{b[1][1][0][0], b[1][1][0][1], b[1][1][0][2], b[1][1][0][3], 0, 0, 0, 0}, cc1, 2, 4, 0, scale1, 0, 127);
            }
            if constexpr (MEAS) { if (A.no_emit == 3) asm volatile("" : : "v"(c0), "v"(c1)); }
            if constexpr (FLOOR >= 2) asm volatile("" : : "v"(c0), "v"(c1));
            else if constexpr (FLOOR == 1) { const uint32_t x0 = or16(c0), x1 = or16(c1); asm volatile("" : : "s"(__builtin_amdgcn_ballot_w64(((x0 | x1) & kPairMask) != 0u))); }
            else if (!(MEAS && A.no_emit == 3)) {
                const uint32_t x0 = or16(c0), x1 = or16(c1);
                if (__builtin_expect(__any(((x0 | x1) & kPairMask) != 0u) && !skip_events, DENSE ? 1 : 0)) {
                    if constexpr (DENSE) dense_event<true>(A, O, c0, c1, L.h[1], pass0 + 64 + r, first_group + 4 * t + 2 * (int32_t) h);
                    else if (half_event<true, MEAS>(A, W, R, c0, c1, L.h[1], x0, x1, pass0 + 64 + r, first_group + 4 * t + 2 * (int32_t) h, low)) { back = t; R.sub = 1u; stop = true; }
                }
            }
            if (!stop) {
                R.sub = 0u;
                if (low) { back = t + 1; stop = true; }
            }
        }
        if (stop) t = n_run;
    }
    if constexpr (NK == 2 && kPfAsm) a_reads_drain(areg);
    R.t = back;
}

// ... and of plain rows (one or two k-blocks)
template <int NK, bool MEAS, bool DENSE, int FLOOR = 0>
__device__ __forceinline__ void f6_class2(const PfArgs &A, MfWave &W, PfOut &O, const char *__restrict__ lds, uint32_t byte_off, int n_row_tiles, int32_t first_group,
                                          uint32_t hw, bool any_n, int64_t pass0, const PfLive2 &L, PfResume &R) {
    static_assert(NK == 1 || NK == 2, "the double pass knows row tiles of one or two k-blocks");
    const uint32_t lane = threadIdx.x & 63u, h = lane >> 5, r = lane & 31u;
    constexpr int kStep = NK * kF6BytesPerKb;
    const char *p = lds + byte_off + lane * (NK == 2 ? 48u : 8u) + (uint32_t) R.t * (uint32_t) kStep;
    // B operands: k-block kb of the window at pass0 + r + 32 o + 64 s covers the bases 16 kb + 8 h ... + 7 behind it: entry r + 8 h + 16 kb + 32 o + 64 s
    i32x4 b[2][2][NK];
    {
        const lds_i32x4 *e = (const lds_i32x4 *) (uintptr_t) (hw + (r + 8u * h) * 16u);
#pragma unroll
        for (int sg = 0; sg < 2; sg++)
#pragma unroll
            for (int o = 0; o < 2; o++) {
#pragma unroll
                for (int kb = 0; kb < NK; kb++) b[sg][o][kb] = e[16 * kb + 32 * o + 64 * sg];
                // rare, wave-uniform: the row tile's last column carries the bias, its base must not read as "no base" (any of the
                // column's four k-slots: they hold the same entry)
                if (any_n && h && ((uint32_t) b[sg][o][NK - 1][3] >> 16) == 0u) b[sg][o][NK - 1][3] |= 0x00020000;
            }
    }
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int n_run = n_row_tiles;
    if constexpr (MEAS) { if (A.no_emit == 2) n_run = 0; }
    if constexpr (FLOOR == 4) n_run = 0;                                             // floor instantiation: the per-pass and per-class set-up alone
    int back = n_row_tiles;
    [[maybe_unused]] uint32_t pa = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) const char *) p;
    [[maybe_unused]] i32x12 areg;
    if constexpr (NK == 2 && kPfAsm) a_reads_begin(pa, areg);
    const bool skip_events = MEAS && A.no_emit >= 1 && A.no_emit <= 3;
    [[maybe_unused]] i32x8 a_fix, a1_fix;                                            // FLOOR 3: ONE operand read per class, the matrix instructions alone in the loop
    if constexpr (FLOOR == 3) { if constexpr (NK == 2) load_a2(pa, a_fix, a1_fix); else { const int2 w0 = *reinterpret_cast<const int2 *>(p), w1 = *reinterpret_cast<const int2 *>(p + 512), w2 = *reinterpret_cast<const int2 *>(p + 1024); a_fix = i32x8{w0.x, w0.y, w1.x, w1.y, w2.x, w2.y, 0, 0}; a1_fix = a_fix; } }
    for (int t = R.t; t < n_run; t++, p += kStep, pa += (uint32_t) kStep) {
        f32x16 c0, c1;
        [[maybe_unused]] i32x8 a, a1;
        bool stop = false, low = false;
        if constexpr (NK == 2 && kPfAsm) plain_product2a_asm(areg, b[0][0][0], b[0][1][0], b[0][0][1], b[0][1][1], c0, c1);
        else if constexpr (NK == 2) { if constexpr (FLOOR == 3) { a = a_fix; a1 = a1_fix; } else load_a2(pa, a, a1); plain_product2_intr(a, a1, b[0][0][0], b[0][1][0], b[0][0][1], b[0][1][1], c0, c1); }
        else {
            if constexpr (FLOOR == 3) a = a_fix;
            else {
                const int2 w0 = *reinterpret_cast<const int2 *>(p), w1 = *reinterpret_cast<const int2 *>(p + 512), w2 = *reinterpret_cast<const int2 *>(p + 1024);
                a = i32x8{w0.x, w0.y, w1.x, w1.y, w2.x, w2.y, 0, 0};
            }
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, i32x8{b[0][0][0][0], b[0][0][0][1], b[0][0][0][2], b[0][0][0][3], 0, 0, 0, 0}, z, 2, 4, 0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, i32x8{b[0][1][0][0], b[0][1][0][1], b[0][1][0][2], b[0][1][0][3], 0, 0, 0, 0}, z, 2, 4, 0, 0, 0, 0);
        }
        if constexpr (MEAS) { if (A.no_emit == 3) asm volatile("" : : "v"(c0), "v"(c1)); }
        if constexpr (FLOOR >= 2) asm volatile("" : : "v"(c0), "v"(c1));
        else if constexpr (FLOOR == 1) { const uint32_t x0 = all_negative(c0), x1 = all_negative(c1); asm volatile("" : : "s"(__builtin_amdgcn_ballot_w64((int) (x0 & x1) >= 0))); }
        else if (R.sub == 0u && !(MEAS && A.no_emit == 3)) {
            const uint32_t x0 = all_negative(c0), x1 = all_negative(c1);
            if (__builtin_expect(__any((int) (x0 & x1) >= 0) && !skip_events, DENSE ? 1 : 0)) {
                if constexpr (DENSE) dense_event<false>(A, O, c0, c1, L.h[0], pass0 + r, first_group + 2 * t + (int32_t) h);
                else if (half_event<false, MEAS>(A, W, R, c0, c1, L.h[0], x0, x1, pass0 + r, first_group + 2 * t + (int32_t) h, low)) { back = t; R.sub = 0u; stop = true; }
            }
        }
        if (!stop) {
            if constexpr (NK == 2 && kPfAsm) plain_product2b_asm<kStep>(pa, areg, b[1][0][0], b[1][1][0], b[1][0][1], b[1][1][1], c0, c1);
            else if constexpr (NK == 2) plain_product2_intr(a, a1, b[1][0][0], b[1][1][0], b[1][0][1], b[1][1][1], c0, c1);
            else {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, i32x8{b[1][0][0][0], b[1][0][0][1], b[1][0][0][2], b[1][0][0][3], 0, 0, 0, 0}, z, 2, 4, 0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, i32x8{b[1][1][0][0], b[1][1][0][1], b[1][1][0][2], b[1][1][0][3], 0, 0, 0, 0}, z, 2, 4, 0, 0, 0, 0);
            }
            if constexpr (MEAS) { if (A.no_emit == 3) asm volatile("" : : "v"(c0), "v"(c1)); }
            if constexpr (FLOOR >= 2) asm volatile("" : : "v"(c0), "v"(c1));
            else if constexpr (FLOOR == 1) { const uint32_t x0 = all_negative(c0), x1 = all_negative(c1); asm volatile("" : : "s"(__builtin_amdgcn_ballot_w64((int) (x0 & x1) >= 0))); }
            else if (!(MEAS && A.no_emit == 3)) {
                const uint32_t x0 = all_negative(c0), x1 = all_negative(c1);
                if (__builtin_expect(__any((int) (x0 & x1) >= 0) && !skip_events, DENSE ? 1 : 0)) {
                    if constexpr (DENSE) dense_event<false>(A, O, c0, c1, L.h[1], pass0 + 64 + r, first_group + 2 * t + (int32_t) h);
                    else if (half_event<false, MEAS>(A, W, R, c0, c1, L.h[1], x0, x1, pass0 + 64 + r, first_group + 2 * t + (int32_t) h, low)) { back = t; R.sub = 1u; stop = true; }
                }
            }
            if (!stop) {
                R.sub = 0u;
                if (low) { back = t + 1; stop = true; }
            }
        }
        if (stop) t = n_run;
    }
    if constexpr (NK == 2 && kPfAsm) a_reads_drain(areg);
    R.t = back;
}

// grid = (blocks per tile, tiles); ONE 1024-thread block per CU (16 waves per CU, <= 128 VGPRs): its waves never meet at a barrier
// after the tables are loaded, so against round 2's two 512-thread blocks the only difference is ONE copy of the tables per CU --
// the other ~60 KB of LDS are the waves' parking space for candidates (and room for larger motif sets in one tile).
// Dynamic LDS: operand tables of the tile | B-operand table (kF6LutBytes) | per-wave sequence staging (kPfStageBytes) | per-wave
// PfEmit (kPfEmitBytes) | per-wave one-hot array of the pass (kPfOnehotBytes) | per-wave parking space (what is left, 16 ... 64 entries per wave).
// Work is handed out per WAVE, without a barrier in the loop: a wave's first unit is its own number, every further unit one
// atomicAdd on one of the tile's kPfCounters counter words (64 bytes apart; the blocks are dealt round-robin onto them and a word
// hands out every kPfCounters-th unit), requested before the current unit is scanned (the atomic's latency hides behind the unit);
// a unit = wave_passes x 64 consecutive window starts, sized on the host (scan_locked).  A block whose CU is still busy with another
// stream's kernel starts late and simply takes fewer units (profiles/r02_stream_coexistence.log, r02_wave_occupancy_ab.log).
// MAXNK: 2 = the kernel for plans whose row tiles all have 1 or 2 k-blocks (motifs of up to 31 columns: every JASPAR-like set);
// 4 = the kernel that also knows row tiles of 3 and 4 k-blocks (its register allocation spills in rare paths).
// MEAS: the measurement-only instantiation (drop candidates, clock stamps); the product kernel carries neither.
// (Measured and dropped in rounds 1-2, tools/pf_variants.py history: fetching the next pass's sequence words early, class
// descriptors in registers, waves walking the classes in rotated order, A operands fetched one row tile ahead (again in round 3, for the
// one-k-block class only, after tools/ubench/insp_probe.hip modes 6 / 7 promised -7 %: +6 % in the kernel), s_setprio around the
// matrix instructions, 12 / 20 / 24 waves per CU, 128 windows per wave, a block-wide hand-out behind barriers, one branch per pair
// of row tiles, a real function call for the rare path.)
// FLOOR (measurement only, MS_MEASURE=1 MS_PF_FLOOR=n; round 6): compile-time cuts of the PRODUCT kernel -- no run-time switch inside, so what is
// left runs exactly as it does in the product -- for the floor table of DESIGN.md section 7: 1 = inspection but no hand-off (no event is ever
// parked), 2 = no inspection either (operand reads + matrix instructions), 3 = the matrix instructions alone (one operand read per class),
// 4 = set-up only (staging, one-hot array, class loop without row tiles).  Their results are void (ms_result::invalid).
template <int MAXNK, bool MEAS, bool DENSE = false, int FLOOR = 0>
__global__ void __launch_bounds__(kPfThreads, 4) prefilter_f6_kernel(const PfArgs A) {
    static_assert(!DENSE || (MAXNK == 2 && !MEAS), "the dense-candidate form exists for the double-pass product kernel");
    static_assert(FLOOR == 0 || (MAXNK == 2 && !MEAS && !DENSE), "the floor instantiations are cuts of the double-pass product kernel");
    extern __shared__ uint4 lds4[];
    constexpr int NT = kPfThreads;
    const TileDesc *__restrict__ T = A.tiles + blockIdx.y;
    const bool wide = MAXNK > 2 && T->max_nk > 2;
    const uint32_t len16 = T->table_len16;
    const uint4 *__restrict__ src = A.tables + T->table_off16;
    for (uint32_t i = threadIdx.x; i < len16; i += NT) lds4[i] = src[i];
    uint4 *lut4 = lds4 + A.lut_off16;
    {
        // byte of four 2-bit codes -> 16 fp4 k-slots (8 bytes): slot 4 c + code_c = 1.0 (e2m1 code 0x2)
        uint2 *lut2 = reinterpret_cast<uint2 *>(lut4);
        for (uint32_t i = threadIdx.x; i < 256u; i += NT) {
            unsigned long long w = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) w |= 2ULL << (4 * (4 * c + (int) ((i >> (2 * c)) & 3u)));
            lut2[i] = make_uint2((uint32_t) w, (uint32_t) (w >> 32));
        }
    }
    // The tile's class descriptors into LDS, 32 bytes apiece behind the B-operand table: a pass reads them from there, the next class's
    // while the current one runs.  (From global memory they came through VECTOR loads, each followed by s_waitcnt vmcnt(0) -- a wait that
    // also covers the next pass's sequence words in flight.  Measured: no difference in time on the benchmark set, profiles/r04_pf_account.log;
    // kept because the pass loop then holds no vector-memory wait but the staging's own.)
    int *cls_lds = reinterpret_cast<int *>(lut4 + 256 * 8 / 16);
    if (threadIdx.x < (uint32_t) kMaxClasses * 8u) {
        const uint32_t ci = threadIdx.x >> 3, f = threadIdx.x & 7u;
        static_assert(sizeof(ClassDesc) == 20, "ClassDesc layout");
        cls_lds[threadIdx.x] = f < 5u ? reinterpret_cast<const int *>(&T->cls[ci])[f] : 0;
    }
    __syncthreads();
    const char *lds = reinterpret_cast<const char *>(lds4);
    const char *lut = reinterpret_cast<const char *>(lut4);
    const int n_classes = T->n_classes;
    // every wave OWNS a first block of the candidate list (no atomic: 4096 waves reserving their first block on one counter word
    // cost 45 us, the whole fixed cost of a small scan); further blocks come from the counter, behind the static ones
    MfWave W;
    W.rq = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) void *) (lds4 + A.rare_off16) + (threadIdx.x >> 6) * (A.rare_cap * (uint32_t) (kRareEntryWords * 4));
    W.rq_cap = A.rare_cap;
    W.rq_flush = A.rare_cap - (A.rare_cap > 32u ? A.rare_cap / 4u : 8u);
    W.rq_n = 0;
    PfEmit *em = reinterpret_cast<PfEmit *>(reinterpret_cast<uint32_t *>(lds4 + A.emit_off16) + (threadIdx.x >> 6) * (uint32_t) kPfEmitWords);
    const uint32_t em_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) void *) em;
    const uint32_t rq_lds = W.rq;
    if ((threadIdx.x & 63u) == 0) {
        em->base = ((unsigned long long) blockIdx.y * gridDim.x + blockIdx.x) * (NT / 64) * A.cand_block + (unsigned long long) (threadIdx.x >> 6) * A.cand_block;
        em->left = A.cand_block;
        em->cand = A.cand; em->n_cand = A.n_cand; em->cand_cap = A.cand_cap; em->cand_static = A.cand_static; em->cand_block = A.cand_block;
    }
    PfOut O;                                                                    // DENSE: the wave's place in the candidate list, in scalar registers
    {
        const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane((int) (((uint32_t) blockIdx.y * gridDim.x + blockIdx.x) * (uint32_t) (NT / 64) + (threadIdx.x >> 6)));
        O.base = (unsigned long long) wave * A.cand_block;
        O.left = A.cand_block;
    }
    const uint32_t lane = threadIdx.x & 63u, r = lane & 31u;
    unsigned long long t0 = 0, r0 = 0;
    unsigned long long cls_cyc[kMaxClasses] = {0, 0, 0, 0, 0, 0};               // measurement only: this wave's cycles inside each class
    if constexpr (MEAS) { if (A.clk) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); } }

    // The sequence words of a pass, staged per wave in LDS: the wave's window starts and the 32 (wide tiles: 64) bases behind the
    // last one span 16 code words and 8 non-ACGT words from pass0 on in a double pass (8 and 4 in a 64-window pass), which lanes
    // 0 ... 23 (11) fetch -- for the NEXT pass, before the current one is scanned, so that the global loads' latency hides behind a
    // pass of matrix work -- and every lane then cuts its own windows (double pass: its entries of the one-hot array) out of the staged words (rounds 1-2: ten global loads per lane and pass, their
    // latency exposed once per pass: a third of the kernel's time on inputs with few row tiles per pass, profiles/r03_c2_latency.log).
    uint32_t *stg = reinterpret_cast<uint32_t *>(lds4 + A.stage_off16) + (threadIdx.x >> 6) * kPfStageWords;
    constexpr bool SH = MAXNK == 2;                                             // the double pass with its one-hot array: kernels without wide classes
    const uint32_t hw_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) void *) (lds4 + A.onehot_off16) + (threadIdx.x >> 6) * (uint32_t) (kOnehotEntries * 16);
    // (32-bit word indices: a set holds <= 2^34 bases = 2^30 code words; the loads then take a scalar base and a 32-bit lane offset)
    const uint32_t n_code_words = (uint32_t) (2 * ((A.n_bases + 31) / 32) + kPadWords), n_mask_words = (uint32_t) ((A.n_bases + 31) / 32 + kPadWords);
    // A pass = PW window starts: 128 in the kernels without wide classes (the double pass: 16 code words and 8 non-ACGT words are staged,
    // 168 bases and their 6 words are needed), 64 in the others (8 + 4 words staged)
    constexpr uint32_t PW = SH ? 128u : 64u, CWP = PW / 16u, NWP = PW / 32u;      // window starts / code words / non-ACGT words per pass
    struct PassWords { uint32_t c, n; };                                        // what lane l loaded: code word l & (2 CWP - 1) and non-ACGT word l & (2 NWP - 1) of the pass
    auto fetch = [&](uint32_t pass) -> PassWords {                              // pass = pass0 / PW
        // every lane loads from both arrays -- the same few cache lines -- through a SCALAR base (the pass is wave-uniform) and a small
        // lane offset: a per-lane 64-bit address costs a register pair that the kernel has not got.  The two results stay two
        // registers until they are staged: choosing between them here would make the wave wait for the loads here
        // (kPadWords >= 16 zero words follow both arrays: only the BASE needs clamping, for the dead passes behind the input's end)
        static_assert(kPadWords >= 16, "the staged words of a pass run up to 16 words past its first");
        const uint32_t bc = pass * CWP < n_code_words - 2u * CWP ? pass * CWP : n_code_words - 2u * CWP, bn = pass * NWP < n_mask_words - 2u * NWP ? pass * NWP : n_mask_words - 2u * NWP;
        typedef const __attribute__((address_space(1))) uint32_t *gptr;          // (a GLOBAL pointer: through an integer it would come back generic)
        auto uniform = [](const uint32_t *q) {                                  // the pointer into scalar registers, whatever the compiler thought of it
            const uint64_t a = reinterpret_cast<uint64_t>(q);
            return (gptr) (((uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) (a >> 32)) << 32) | (uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) a));
        };
        return PassWords{uniform(A.codes + bc)[lane & (2u * CWP - 1u)], uniform(A.nmask + bn)[lane & (2u * NWP - 1u)]};
    };
    auto scan_pass = [&](int64_t pass0) {                                    // 64 window starts of this wave (pass0 ... + 63, wave-uniform) against every class
        // (32-bit: what is left of the input from pass0 on is wave-uniform; a 64-bit position per lane costs two register pairs)
        const int64_t left64 = A.n_bases - pass0;
        const uint32_t left = left64 >= 64 ? 64u : (left64 > 0 ? (uint32_t) left64 : 0u);
        bool live0 = r < left, live1 = r + 32u < left;
        PassSeq Q;
        Q.stg = stg;
        Q.cw[0] = staged_cw(stg, r, 0);
        Q.cw[1] = staged_cw(stg, r, 1);
        const uint32_t nw0 = staged_nw(stg, r, 0), nw1 = staged_nw(stg, r, 1);
        const uint32_t nw2 = wide ? staged_nw(stg, r, 2) : 0u;                           // only classes of 3 or 4 k-blocks reach bases 64 ... 95
        Q.any_n = __any((nw0 | nw1 | nw2) != 0u);
        if (Q.any_n && A.skip_alln) {
            // a window whose bases are ALL non-ACGT (the tile's motifs span <= 32 bases, <= 64 with wide classes) scores 0 on every
            // motif and none reports that (plan: every threshold > 0): such lanes queue nothing, and a pass made of them only --
            // the inside of an assembly gap -- is skipped whole
            const bool dead0 = nw0 == 0xFFFFFFFFu && (!wide || nw1 == 0xFFFFFFFFu);
            const bool dead1 = nw1 == 0xFFFFFFFFu && (!wide || nw2 == 0xFFFFFFFFu);
            live0 = live0 && !dead0;
            live1 = live1 && !dead1;
            if (!__any(live0 || live1)) return;
        }
        const PfLive L{live0, live1, __builtin_amdgcn_ballot_w64(live0), __builtin_amdgcn_ballot_w64(live1)};
        auto read_cd = [&](int i) { return *reinterpret_cast<const int4 *>(cls_lds + 8 * i); };      // {nk, n_row_tiles, base16, first_group}; paired: word 4
        int4 cd4 = read_cd(0);
        int cdp = cls_lds[4];
        for (int i = 0; i < n_classes; i++) {
            ClassDesc cd;                                                         // wave-uniform: into scalar registers
            cd.nk = __builtin_amdgcn_readfirstlane(cd4.x);
            cd.n_row_tiles = __builtin_amdgcn_readfirstlane(cd4.y);
            cd.base16 = (uint32_t) __builtin_amdgcn_readfirstlane(cd4.z);
            cd.first_group = __builtin_amdgcn_readfirstlane(cd4.w);
            cd.paired = __builtin_amdgcn_readfirstlane(cdp);
            if (i + 1 < n_classes) { cd4 = read_cd(i + 1); cdp = cls_lds[8 * (i + 1) + 4]; }      // the next class's, while this one runs
            const uint32_t off = cd.base16 * 16u;
            PfResume R{0, 0u, 0u};
            unsigned long long tc0 = 0;
            if constexpr (MEAS) { if (A.cls_clk) tc0 = __builtin_amdgcn_s_memtime(); }
            while (R.t < cd.n_row_tiles) {                                        // a class comes back early when the parking space runs low
                if (cd.paired) {
                    if (cd.nk == 1) f6_pair_class<1, MEAS>(A, W, lds, lut, off, cd.n_row_tiles, cd.first_group, Q, pass0, L, R);
                    else f6_pair_class<2, MEAS>(A, W, lds, lut, off, cd.n_row_tiles, cd.first_group, Q, pass0, L, R);
                } else {
                    switch (cd.nk) {
                        case 1: f6_class<1, MEAS>(A, W, lds, lut, off, cd.n_row_tiles, cd.first_group, Q, pass0, L, R); break;
                        case 2: f6_class<2, MEAS>(A, W, lds, lut, off, cd.n_row_tiles, cd.first_group, Q, pass0, L, R); break;
                        case 3: if constexpr (MAXNK > 2) f6_class<3, MEAS>(A, W, lds, lut, off, cd.n_row_tiles, cd.first_group, Q, pass0, L, R); else R.t = cd.n_row_tiles; break;
                        case 4: if constexpr (MAXNK > 2) f6_class<4, MEAS>(A, W, lds, lut, off, cd.n_row_tiles, cd.first_group, Q, pass0, L, R); else R.t = cd.n_row_tiles; break;
                        default: R.t = cd.n_row_tiles; break;
                    }
                }
                if (W.rq_n >= W.rq_flush) { pf_flush(em_lds, rq_lds, W.rq_n); W.rq_n = 0; }
            }
            if constexpr (MEAS) { if (A.cls_clk) cls_cyc[i] += __builtin_amdgcn_s_memtime() - tc0; }
        }
    };
    auto scan_pass2 = [&](int64_t pass0, bool any_n) {                        // the double pass: 128 window starts of this wave (pass0 ... + 127, wave-uniform) against every class
        const int64_t left64 = A.n_bases - pass0;
        const uint32_t left = left64 >= 128 ? 128u : (left64 > 0 ? (uint32_t) left64 : 0u);
        bool lv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) lv[k] = r + 32u * (uint32_t) k < left;
        const uint32_t *stn = stg + 16;                                          // the staged non-ACGT words
        {
            // the pass's one-hot array (above PassSeq's helpers): lane l makes entries l, 64 + l and 128 + (l & 31) -- the 8 bases from pass0 + entry on,
            // cut out of the staged words with funnel shifts; a non-ACGT base is an all-zero column (any_n: rare, wave-uniform)
            auto entry = [&](uint32_t x) {
                const uint32_t w = x >> 4;
                i32x8 e = onehot_f4(lut, __builtin_amdgcn_alignbit(stg[w + 1], stg[w], (x & 15u) * 2u) & 0xFFFFu);
                if (any_n) clear_n(e, __builtin_amdgcn_alignbit(stn[(x >> 5) + 1], stn[x >> 5], x & 31u) & 0xFFu);
                return i32x4{e[0], e[1], e[2], e[3]};
            };
            lds_i32x4 *hq = (lds_i32x4 *) (uintptr_t) hw_lds;
            hq[lane] = entry(lane);
            hq[64u + lane] = entry(64u + lane);
            hq[128u + r] = entry(128u + r);
        }
        if (any_n && A.skip_alln) {
            // a window whose bases are ALL non-ACGT (the tile's motifs span <= 32 bases) scores 0 on every motif and none reports that (plan:
            // every threshold > 0): such lanes queue nothing, and a pass made of them only -- the inside of an assembly gap -- is skipped whole
#pragma unroll
            for (int k = 0; k < 4; k++) lv[k] = lv[k] && __builtin_amdgcn_alignbit(stn[k + 1], stn[k], r) != 0xFFFFFFFFu;
            if (!__any(lv[0] || lv[1] || lv[2] || lv[3])) return;
        }
        const PfLive2 L{{PfLive{lv[0], lv[1], __builtin_amdgcn_ballot_w64(lv[0]), __builtin_amdgcn_ballot_w64(lv[1])},
                         PfLive{lv[2], lv[3], __builtin_amdgcn_ballot_w64(lv[2]), __builtin_amdgcn_ballot_w64(lv[3])}}};
        auto read_cd = [&](int i) { return *reinterpret_cast<const int4 *>(cls_lds + 8 * i); };      // {nk, n_row_tiles, base16, first_group}; paired: word 4
        int4 cd4 = read_cd(0);
        int cdp = cls_lds[4];
        for (int i = 0; i < n_classes; i++) {
            ClassDesc cd;                                                         // wave-uniform: into scalar registers
            cd.nk = __builtin_amdgcn_readfirstlane(cd4.x);
            cd.n_row_tiles = __builtin_amdgcn_readfirstlane(cd4.y);
            cd.base16 = (uint32_t) __builtin_amdgcn_readfirstlane(cd4.z);
            cd.first_group = __builtin_amdgcn_readfirstlane(cd4.w);
            cd.paired = __builtin_amdgcn_readfirstlane(cdp);
            if (i + 1 < n_classes) { cd4 = read_cd(i + 1); cdp = cls_lds[8 * (i + 1) + 4]; }      // the next class's, while this one runs
            const uint32_t off = cd.base16 * 16u;
            PfResume R{0, 0u, 0u, 0u};
            unsigned long long tc0 = 0;
            if constexpr (MEAS) { if (A.cls_clk) tc0 = __builtin_amdgcn_s_memtime(); }
            while (R.t < cd.n_row_tiles) {                                        // a class comes back early when the parking space runs low
                if (cd.paired) {
                    if (cd.nk == 1) f6_pair_class2<1, MEAS, DENSE, FLOOR>(A, W, O, lds, off, cd.n_row_tiles, cd.first_group, hw_lds, pass0, L, R);
                    else f6_pair_class2<2, MEAS, DENSE, FLOOR>(A, W, O, lds, off, cd.n_row_tiles, cd.first_group, hw_lds, pass0, L, R);
                } else if (cd.nk == 1) f6_class2<1, MEAS, DENSE, FLOOR>(A, W, O, lds, off, cd.n_row_tiles, cd.first_group, hw_lds, any_n, pass0, L, R);
                else if (cd.nk == 2) f6_class2<2, MEAS, DENSE, FLOOR>(A, W, O, lds, off, cd.n_row_tiles, cd.first_group, hw_lds, any_n, pass0, L, R);
                else R.t = cd.n_row_tiles;
                if (W.rq_n >= W.rq_flush) { pf_flush(em_lds, rq_lds, W.rq_n); W.rq_n = 0; }
            }
            if constexpr (MEAS) { if (A.cls_clk) cls_cyc[i] += __builtin_amdgcn_s_memtime() - tc0; }
        }
    };
    {
        const uint32_t wave_passes = A.wave_passes < 1 ? 8u : (uint32_t) A.wave_passes;
        const uint32_t n_passes_total = (uint32_t) ((A.n_bases + PW - 1) / PW);       // <= 2^28: a set holds <= 2^34 bases
        const uint32_t n_units = (n_passes_total + wave_passes - 1) / wave_passes;
        constexpr uint32_t wpb = NT / 64;
        // A small input (A.use_counters == 0): one unit per wave, no atomic at all.  Else kPfCounters counter words per tile: a
        // wave's first unit in its word's group is its own number there; the words start at 0 and the waves add their group's
        // size themselves.
        const bool dyn = A.use_counters != 0;
        const uint32_t K = !dyn ? 1u : (gridDim.x < (uint32_t) kPfCounters ? gridDim.x : (uint32_t) kPfCounters);     // every word needs a block
        const uint32_t g = blockIdx.x % K;
        const uint32_t waves_g = ((gridDim.x - g + K - 1) / K) * wpb;
        const uint32_t units_g = n_units > g ? (n_units - g + K - 1) / K : 0u;
        unsigned int *word = A.chunk_counter + ((size_t) blockIdx.y * kPfCounters + g) * 16;
        uint32_t v = (uint32_t) __builtin_amdgcn_readfirstlane((int) ((blockIdx.x / K) * wpb + (threadIdx.x >> 6)));     // wave-uniform: everything derived from it lives in scalar registers
        PassWords words = v < units_g ? fetch((v * K + g) * wave_passes) : PassWords{0u, 0u};
        while (v < units_g) {
            uint32_t next = 0xFFFFFFFFu;
            // the next unit: asked for before this one is scanned, looked at in its last pass.  The instruction by name: hipcc's atomicAdd is
            // followed at once by s_waitcnt vmcnt(0) (its wave-aggregated form broadcasts the result), which exposed the counter's round
            // trip once per unit
            unsigned int u = 0;
            const uint32_t uid = v * K + g;                                       // the unit: window starts [uid, uid + 1) * PW * wave_passes
            const uint32_t p0 = uid * wave_passes;
            for (uint32_t j = 0; j < wave_passes; j++) {                          // passes past the end scan dead lanes (last unit only)
                if (lane < 3u * CWP) stg[lane] = lane < 2u * CWP ? words.c : words.n;       // (the wave's LDS operations execute in order: no barrier)
                [[maybe_unused]] const bool pass_any_n = SH && __any((lane & (2u * NWP - 1u)) < 6u && words.n != 0u);      // (double pass: a non-ACGT base among the 192 staged)
                // (behind the staging, which waits for every vector-memory operation in flight)
                // INVARIANT: the compiler believes `u` is ready at once, the value arrives with the atomic's return.  Nothing may read,
                // copy or spill u's register before a vmcnt(0) wait has retired the atomic: with wave_passes >= 2 that is pass 1's staging wait
                // (every later copy is then harmless), and pass 0's scan must not touch the register -- checked on the built code object by
                // tests/test_host_cabi.py::test_prefilter_isa_resources_and_the_atomic_register.  The host never launches wave_passes == 1 with
                // the counters on (scan_locked refuses it); if it ever did, the compiler's own atomicAdd (waited for at once) takes over.
                if (j == 0 && dyn && lane == 0) {
                    if (kPfAsm && wave_passes >= 2) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(u) : "v"(word), "v"(1u) : "memory");
                    else u = atomicAdd(word, 1u);                                  // (the default build: the compiler's atomic, waited for where it is used)
                }
                // the next pass's words -- of this unit, or the first of the wave's NEXT unit (its number arrived long ago) -- are in
                // flight while this pass is scanned
                if (j + 1 < wave_passes) words = fetch(p0 + j + 1);
                else {
                    if (dyn) {
                        if constexpr (kPfAsm) asm volatile("s_waitcnt vmcnt(0)" : "+v"(u) : : "memory");     // (the compiler does not know the atomic is in flight)
                        next = waves_g + (uint32_t) __builtin_amdgcn_readfirstlane((int) u);
                    }
                    if (next < units_g) words = fetch((next * K + g) * wave_passes);
                }
                if constexpr (SH) scan_pass2((int64_t) (p0 + j) * 128, pass_any_n);
                else scan_pass((int64_t) (p0 + j) * 64);
            }
            v = next;
        }
    }
    if (W.rq_n) pf_flush(em_lds, rq_lds, W.rq_n);
    {                                                                             // the unused rest of the last block: empty records
        const unsigned long long base = DENSE ? O.base : em->base;
        const uint32_t left = DENSE ? O.left : em->left;
        for (uint32_t i = 0; i < left; i += 64) {
            const unsigned long long j = base + i + lane;
            if (i + lane < left && j < A.cand_cap) A.cand[j] = 0ULL;
        }
    }
    if constexpr (MEAS) {
        if (A.clk && threadIdx.x == 0) {
            const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
            const size_t b = (size_t) blockIdx.y * gridDim.x + blockIdx.x;
            A.clk[kPfClkWords * b] = t1 - t0;
            A.clk[kPfClkWords * b + 1] = r1 - r0;
            for (int i = 0; i < kMaxClasses; i++) A.clk[kPfClkWords * b + 2 + i] = cls_cyc[i];
        }
    }
}

// -------------------------------------------------------------------- fp64 kernels --

// grid = (ceil(n_bases/256), n_exact motifs).  Fallback for motifs the pre-filter cannot take
// (W > 63, max_raw <= 0, non-finite values, a cutoff so low that (almost) every window passes).
__global__ void __launch_bounds__(256) exact_all_kernel(const DevSeq S, const DevPwm Pw, const int32_t *__restrict__ motifs,
                                                        int strand_mask, const HitOut H) {
    const int64_t g = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= S.n_bases) return;
    const int32_t p = motifs[blockIdx.y];
    const int W = Pw.width[p];
    const int64_t r = find_region(S, g);
    if (g + W > S.offsets[r + 1]) return;
    double fwd, rev;
    score_window(S, Pw.tab2 + Pw.tab_off[p], W, g, fwd, rev);
    test_and_emit(H, Pw, (uint32_t) p, hit_coord(H, S, r, g), fwd, rev, strand_mask);
}

// The same with the motif's table in LDS (round 6): exact_all_kernel reads a table entry per (window, column) through the
// texture addressers -- the unit the fp64 stage is bound by (profiles/r03t_rescore_ta.log) -- although a wave's 64 lanes want at most four
// different entries of a column.  Here a block stages its motif's W x 4 entries (+ an all-zero entry for the columns that add nothing) once,
// scans kExactIter strips of 256 window starts against them, and a column is one 16-byte LDS read (a broadcast: <= 5 distinct addresses per
// wave) + the two fp64 adds, in the reference's column order; a column that adds nothing ADDS the zero entry (score_window32's argument: a
// running sum that started at +0.0 is never -0.0, so x + (+0.0) = x bit for bit).  Motifs wider than kExactTileMaxW keep exact_all_kernel.
constexpr int kExactTileMaxW = 1024;       // 64 KB of LDS for the table
constexpr int kExactIter = 8;

__global__ void __launch_bounds__(256) exact_tiled_kernel(const DevSeq S, const DevPwm Pw, const int32_t *__restrict__ motifs, int strand_mask, const HitOut H) {
    extern __shared__ double2 tab_lds[];                       // [W * 4 + 1]
    const int32_t p = motifs[blockIdx.y];
    const int W = Pw.width[p];
    {
        const double2 *__restrict__ tab = Pw.tab2 + Pw.tab_off[p];
        for (int i = threadIdx.x; i < W * 4; i += 256) tab_lds[i] = tab[i];
        if (threadIdx.x == 0) tab_lds[W * 4] = make_double2(0.0, 0.0);
    }
    __syncthreads();
    const uint32_t zero = (uint32_t) W * 4u;
    for (int it = 0; it < kExactIter; it++) {
        const int64_t g = ((int64_t) blockIdx.x * kExactIter + it) * 256 + threadIdx.x;
        if (g >= S.n_bases) break;
        const int64_t r = find_region(S, g);
        if (g + W > S.offsets[r + 1]) continue;
        double fwd = 0.0, rev = 0.0;
        for (int c0 = 0; c0 < W; c0 += 32) {
            const uint64_t cw = code_window(S.codes, g + c0);
            const int n = (W - c0) < 32 ? (W - c0) : 32;
            const uint32_t skip = n_window(S.nmask, g + c0) | ~low_mask(n);           // bit c: column c0 + c adds nothing
            for (int c1 = 0; c1 < n; c1 += 8) {
                double2 t[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int c = c1 + k;
                    const uint32_t idx = (uint32_t) (c0 + c) * 4u + ((uint32_t) (cw >> (2 * c)) & 3u);
                    t[k] = tab_lds[((skip >> c) & 1u) ? zero : idx];
                }
#pragma unroll
                for (int k = 0; k < 8; k++) { fwd += t[k].x; rev += t[k].y; }
            }
        }
        test_and_emit(H, Pw, (uint32_t) p, hit_coord(H, S, r, g), fwd, rev, strand_mask);
    }
}

// U candidate records per thread and round, their loads issued side by side: the kernel is a chain of dependent gathers
// (record -> region hint / sequence words / motif id -> offsets / width / table offset -> table entries), so the records in
// flight per thread -- not the arithmetic -- set its speed; one barrier pair per round of U records instead of per record.
constexpr int kRescoreU = 4;

__global__ void __launch_bounds__(256) rescore_kernel(const DevSeq S, const DevPwm Pw, const uint64_t *__restrict__ cand,
                                                      const unsigned long long *__restrict__ n_cand, uint64_t n_static, uint64_t cand_cap,
                                                      const FieldMeta *__restrict__ field_meta, int strand_mask,
                                                      const HitOut H) {
    __shared__ HitStage st;
    if (threadIdx.x == 0) st.n = 0;
    __syncthreads();
    unsigned long long n = n_static + *n_cand;            // the waves' own first blocks, then the blocks they reserved from the counter
    if (n > cand_cap) n = cand_cap;
    constexpr int U = kRescoreU;
    const bool both = strand_mask == 3;                  // both strands: fields 2k, 2k + 1 = motif slot k forward, reverse; one strand: field n = slot n
    const unsigned long long per_sub = (unsigned long long) gridDim.x * blockDim.x;
    const unsigned long long per_round = per_sub * U;
    const unsigned long long rounds = (n + per_round - 1) / per_round;
    const int4 *__restrict__ meta4 = reinterpret_cast<const int4 *>(field_meta);
    for (unsigned long long rd = 0; rd < rounds; rd++) {
        const unsigned long long i0 = rd * per_round + (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
        uint64_t c[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            live[u] = i0 + u * per_sub < n;
            c[u] = live[u] ? cand[i0 + u * per_sub] : 0;
        }
        // independent of each other: the region's place (one read), sequence words, N words, the first flagged field's motif / width / table (one read)
        int64_t g[U];
        int4 bi[U];
        uint64_t cw[U];
        uint32_t nw[U], flags[U];
        int32_t group[U];
        int4 fm[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            g[u] = (int64_t) (c[u] >> 30);
            group[u] = (int32_t) ((c[u] >> 16) & 0x3FFFu);
            const uint32_t f = (uint32_t) c[u] & 0xFFFFu;              // bit n = field n
            flags[u] = both ? (f | (f >> 1)) & 0x5555u : f;             // a motif's two strands are re-scored together anyway
            bi[u] = S.blkinfo[g[u] >> 6];
            cw[u] = code_window(S.codes, g[u]);
            nw[u] = n_window(S.nmask, g[u]);
            fm[u] = flags[u] ? meta4[group[u] * kGroupFields + (__ffs((int) flags[u]) - 1)] : make_int4(-1, 0, 0, 0);
        }
        // the region's bounds: out of the block's record; only tiny regions (a third region start within the block's reach) and
        // starts beyond 32 bits need the offsets themselves
        int64_t r[U], beg[U], end[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t base = g[u] & ~(int64_t) 63;
            int64_t lo = bi[u].x, o0 = base + bi[u].y, o1 = base + bi[u].z, o2 = base + bi[u].w;
            if (bi[u].x < 0) {
                lo = S.blk2reg[g[u] >> 6];
                o0 = S.offsets[lo]; o1 = S.offsets[lo + 1];
                o2 = lo + 2 <= S.R ? S.offsets[lo + 2] : o1;
            }
            if (g[u] < o1) { r[u] = lo; beg[u] = o0; end[u] = o1; }
            else if (g[u] < o2) { r[u] = lo + 1; beg[u] = o1; end[u] = o2; }
            else { r[u] = find_region(S, g[u]); beg[u] = S.offsets[r[u]]; end[u] = S.offsets[r[u] + 1]; }      // tiny regions
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (!live[u]) continue;
            const int64_t gk = H.pbits ? (int64_t) (((uint64_t) r[u] << H.pbits) | (uint64_t) (g[u] - beg[u])) : g[u];
            bool first = true;
            while (flags[u]) {
                const int field = __ffs((int) flags[u]) - 1;
                flags[u] &= flags[u] - 1u;
                int4 f4 = fm[u];
                if (!first) f4 = meta4[group[u] * kGroupFields + field];           // further motifs of the group: rare
                first = false;
                const int32_t m = f4.x;
                const int w = f4.y;
                if (m < 0) continue;
                if (g[u] + w > end[u]) continue;                         // window runs past its region (cscore.c:340)
                double fwd, rev;
                if (w <= 32 && Pw.tab32) score_window32(Pw.tab2, (uint32_t) f4.z, Pw.zero_bytes, w, cw[u], nw[u], fwd, rev);     // non-ACGT bases add nothing (cscore.c:345-353)
                else score_window(S, Pw.tab2 + Pw.tab_off[m], w, g[u], fwd, rev);
                test_and_stage(st, H, Pw, (uint32_t) m, gk, fwd, rev, strand_mask, __int_as_float(f4.w));
            }
        }
        __syncthreads();
        const bool full = st.n > (unsigned int) (kHitStage - 256 * 2 * U);
        __syncthreads();                     // every thread has read st.n before any wave can append again: the decision is block-uniform
        if (full) stage_flush(st, H);
    }
    stage_flush(st, H);
}

// ----------------------------------------------------------------------- finalize --

// n_dev != nullptr: the number of hits is only known on the device (a scan whose sizes were predicted, scan_locked): n is then
// the launch's capacity and the true count min(*n_dev, n) is read here.
__global__ void __launch_bounds__(256) finalize_kernel(const uint64_t *__restrict__ keys, int64_t n, const unsigned long long *__restrict__ n_dev,
                                                       int gbits, int32_t P, const DevSeq S,
                                                       int64_t *__restrict__ seq_idx, int64_t *__restrict__ pos,
                                                       int8_t *__restrict__ strand, int64_t *__restrict__ motif_first,
                                                       unsigned long long *__restrict__ region_counts) {
    if (n_dev) { const unsigned long long nd = *n_dev; if ((unsigned long long) n > nd) n = (int64_t) nd; }
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    uint32_t motif = 0xFFFFFFFFu;
    bool new_pair = false;
    if (live) {
        const uint64_t k = keys[i];
        const uint64_t gmask = (1ULL << gbits) - 1ULL;
        const int64_t g = (int64_t) ((k >> 1) & gmask);
        motif = (uint32_t) (k >> (gbits + 1));
        const int64_t r = find_region(S, g);
        seq_idx[i] = r;
        pos[i] = g - S.offsets[r];
        strand[i] = (int8_t) ((k & 1ULL) ? 2 : 1);
        bool first_of_motif = (i == 0);
        new_pair = true;
        int64_t q0 = 0;                                  // per-motif offsets: every motif after the previous hit's up to this one starts here
        if (i > 0) {
            const uint64_t kp = keys[i - 1];
            const uint32_t mp = (uint32_t) (kp >> (gbits + 1));
            first_of_motif = mp != motif;
            q0 = (int64_t) mp + 1;
            if (!first_of_motif) {
                const int64_t gp = (int64_t) ((kp >> 1) & gmask);
                new_pair = gp < S.offsets[r];            // previous hit of this motif lies in an earlier region
            }
        }
        if (first_of_motif) for (int64_t q = q0; q <= (int64_t) motif; q++) motif_first[q] = i;
        if (i == n - 1) for (int64_t q = (int64_t) motif + 1; q <= P; q++) motif_first[q] = n;     // motifs after the last hit: empty
    }
    // number of regions with >= 1 hit per motif (stats.py:29-31): one atomic per (wave, motif)
    unsigned long long todo = __ballot(live && new_pair);
    while (todo) {
        const int leader = __ffsll((long long) todo) - 1;
        const uint32_t m = __shfl(motif, leader);
        const unsigned long long same = __ballot(live && new_pair && motif == m);
        if ((int) (threadIdx.x & 63) == leader) atomicAdd(&region_counts[m], (unsigned long long) __popcll(same));
        todo &= ~same;
    }
}

// The same when the keys carry (region, position inside the region): nothing to look up, only bits to unpack.
// Four consecutive hits per thread: 16-byte loads and stores, the four strand bytes as one word.
__global__ void __launch_bounds__(256) finalize_rp_kernel(const uint64_t *__restrict__ keys, int64_t n, const unsigned long long *__restrict__ n_dev,
                                                          int rbits, int pbits, int32_t P,
                                                          int64_t *__restrict__ seq_idx, int64_t *__restrict__ pos,
                                                          int8_t *__restrict__ strand, int64_t *__restrict__ motif_first,
                                                          unsigned long long *__restrict__ region_counts) {
    if (n_dev) { const unsigned long long nd = *n_dev; if ((unsigned long long) n > nd) n = (int64_t) nd; }
    const int64_t i0 = 4 * ((int64_t) blockIdx.x * blockDim.x + threadIdx.x);
    const bool live = i0 < n;
    uint32_t motif0 = 0xFFFFFFFFu;
    int n_new = 0;                                              // new (motif, region) pairs among this thread's hits of motif0
    if (live) {
        uint64_t k[4];
        const bool full = i0 + 4 <= n;
        if (full) {
            const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(keys + i0), b2 = *reinterpret_cast<const ulonglong2 *>(keys + i0 + 2);
            k[0] = a.x; k[1] = a.y; k[2] = b2.x; k[3] = b2.y;
        } else {
            for (int j = 0; j < 4; j++) k[j] = i0 + j < n ? keys[i0 + j] : 0;
        }
        uint64_t prev = i0 > 0 ? keys[i0 - 1] >> (pbits + 1) : ~0ULL;
        int64_t sq[4], ps[4];
        uint32_t sd = 0;
        const uint64_t rmask = (1ULL << rbits) - 1ULL, pmask = (1ULL << pbits) - 1ULL;
        motif0 = (uint32_t) (k[0] >> (pbits + 1 + rbits));
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (i0 + j < n) {
                const uint64_t pair = k[j] >> (pbits + 1);      // (motif, region)
                const uint32_t motif = (uint32_t) (pair >> rbits);
                sq[j] = (int64_t) (pair & rmask);
                ps[j] = (int64_t) ((k[j] >> 1) & pmask);
                sd |= ((k[j] & 1ULL) ? 2u : 1u) << (8 * j);
                if (prev == ~0ULL || (uint32_t) (prev >> rbits) != motif)        // every motif after the previous hit's up to this one starts here
                    for (int64_t q = prev == ~0ULL ? 0 : (int64_t) (uint32_t) (prev >> rbits) + 1; q <= (int64_t) motif; q++) motif_first[q] = i0 + j;
                if (i0 + j == n - 1) for (int64_t q = (int64_t) motif + 1; q <= P; q++) motif_first[q] = n;   // motifs after the last hit: empty
                if (prev != pair) {
                    if (motif == motif0) n_new++;
                    else atomicAdd(&region_counts[motif], 1ULL);    // a thread's hits rarely span two motifs
                }
                prev = pair;
            }
        }
        if (full) {
            *reinterpret_cast<longlong2 *>(seq_idx + i0) = make_longlong2(sq[0], sq[1]);
            *reinterpret_cast<longlong2 *>(seq_idx + i0 + 2) = make_longlong2(sq[2], sq[3]);
            *reinterpret_cast<longlong2 *>(pos + i0) = make_longlong2(ps[0], ps[1]);
            *reinterpret_cast<longlong2 *>(pos + i0 + 2) = make_longlong2(ps[2], ps[3]);
            *reinterpret_cast<uint32_t *>(strand + i0) = sd;
        } else {
            for (int j = 0; j < 4 && i0 + j < n; j++) { seq_idx[i0 + j] = sq[j]; pos[i0 + j] = ps[j]; strand[i0 + j] = (int8_t) ((sd >> (8 * j)) & 0xFFu); }
        }
    }
    // regions with >= 1 hit per motif (stats.py:29-31): one atomic per (wave, motif)
    unsigned long long todo = __ballot(live && n_new > 0);
    while (todo) {
        const int leader = __ffsll((long long) todo) - 1;
        const uint32_t m = __shfl(motif0, leader);
        const unsigned long long same = __ballot(live && n_new > 0 && motif0 == m);
        int v = (live && motif0 == m) ? n_new : 0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if ((int) (threadIdx.x & 63) == leader) atomicAdd(&region_counts[m], (unsigned long long) v);
        todo &= ~same;
    }
}

// -------------------------------------------------------------- de-dup / site tables --

__device__ __forceinline__ int32_t motif_of_hit(const int64_t *__restrict__ motif_off, int32_t P, int64_t i) {
    int32_t lo = 0, hi = P;                       // motif_off[lo] <= i < motif_off[hi]
    while (hi - lo > 1) {
        const int32_t mid = (lo + hi) >> 1;
        if (motif_off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// scanner.py:156-193 on the sorted hit arrays.  The thread of the FIRST hit of a (motif, region)
// segment walks the segment once with one "current site" per strand: a later same-strand site
// closer than the motif width either loses (score <=: tie keeps the earlier one) or replaces it.
// The kept hits are already in the order the reference returns (start ascending, '+' first).
__global__ void __launch_bounds__(256) dedup_kernel(int64_t n, const int64_t *__restrict__ motif_off, int32_t P,
                                                    const int32_t *__restrict__ width, const int64_t *__restrict__ seq_idx,
                                                    const int64_t *__restrict__ pos, const double *__restrict__ score,
                                                    const int8_t *__restrict__ strand, uint32_t *__restrict__ keep) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t p = motif_of_hit(motif_off, P, i);
    const int64_t r = seq_idx[i];
    if (i > motif_off[p] && seq_idx[i - 1] == r) return;            // not the head of its segment
    const int64_t end = motif_off[p + 1];
    const int64_t W = width[p];
    int64_t cur[2] = {-1, -1};
    for (int64_t j = i; j < end && seq_idx[j] == r; j++) {
        const int s = strand[j] == 1 ? 0 : 1;
        uint32_t kj = 1;
        if (cur[s] >= 0 && pos[j] - pos[cur[s]] < W) {
            if (score[cur[s]] >= score[j]) kj = 0;                   // scanner.py:163-164
            else { keep[cur[s]] = 0; cur[s] = j; }                   // scanner.py:165-166
        } else {
            cur[s] = j;
        }
        keep[j] = kj;
    }
}

__global__ void __launch_bounds__(256) compact_hits_kernel(int64_t n, const uint32_t *__restrict__ keep,
                                                           const uint64_t *__restrict__ dst,
                                                           const int64_t *__restrict__ seq_in, const int64_t *__restrict__ pos_in,
                                                           const double *__restrict__ score_in, const int8_t *__restrict__ strand_in,
                                                           int64_t *__restrict__ seq_out, int64_t *__restrict__ pos_out,
                                                           double *__restrict__ score_out, int8_t *__restrict__ strand_out) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i]) return;
    const uint64_t d = dst[i];
    seq_out[d] = seq_in[i]; pos_out[d] = pos_in[i]; score_out[d] = score_in[i]; strand_out[d] = strand_in[i];
}

// new per-motif offsets after compaction: off_out[p] = dst[off_in[p]] (or the kept total at the end)
__global__ void remap_offsets_kernel(const int64_t *__restrict__ off_in, int32_t P, int64_t n, const uint64_t *__restrict__ dst,
                                     const uint32_t *__restrict__ keep, int64_t *__restrict__ off_out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > P) return;
    const int64_t o = off_in[p];
    off_out[p] = o < n ? (int64_t) dst[o] : (n > 0 ? (int64_t) dst[n - 1] + keep[n - 1] : 0);
}

// io/__init__.py:23-33: per (motif, region) the number of sites and the maximum score
__global__ void __launch_bounds__(256) site_tables_kernel(int64_t n, const int64_t *__restrict__ motif_off, int32_t P, int64_t R,
                                                          const int64_t *__restrict__ seq_idx, const double *__restrict__ score,
                                                          int32_t *__restrict__ n_sites, double *__restrict__ max_score) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t p = motif_of_hit(motif_off, P, i);
    const int64_t r = seq_idx[i];
    if (i > motif_off[p] && seq_idx[i - 1] == r) return;
    const int64_t end = motif_off[p + 1];
    int32_t cnt = 0;
    double best = score[i];
    for (int64_t j = i; j < end && seq_idx[j] == r; j++) {
        cnt++;
        if (score[j] > best) best = score[j];
    }
    n_sites[(int64_t) p * R + r] = cnt;
    max_score[(int64_t) p * R + r] = best;
}

// ---------------------------------------------------------------- window sweep (N3) --
// A sweep scans one chromosome span as ONE region and hands every hit to each window that holds it whole:
// window k = [k * stride, k * stride + window) of the span; a hit of a width-W motif at span position g lies in
// windows ceil((g + W - window) / stride) .. floor(g / stride)  (cscore.c:340: the window must contain all W bases).
__device__ __forceinline__ void sweep_window_range(int64_t g, int W, int32_t window, int32_t stride, int64_t n_windows,
                                                   int64_t &lo, int64_t &hi) {
    hi = g / stride;
    if (hi > n_windows - 1) hi = n_windows - 1;
    const int64_t need = g + W - window;                       // smallest window start that still holds the site
    lo = need <= 0 ? 0 : (need + stride - 1) / stride;
}

__global__ void __launch_bounds__(256) sweep_count_kernel(int64_t n, const int64_t *__restrict__ motif_off, int32_t P,
                                                          const int32_t *__restrict__ width, const int64_t *__restrict__ pos,
                                                          int32_t window, int32_t stride, int64_t n_windows,
                                                          uint32_t *__restrict__ cnt) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t m = motif_of_hit(motif_off, P, i);
    int64_t lo, hi;
    sweep_window_range(pos[i], width[m], window, stride, n_windows, lo, hi);
    cnt[i] = hi >= lo ? (uint32_t) (hi - lo + 1) : 0u;
}

// Hand-out without a sort.  Within a motif the hits are ordered by span position g (then strand), and both ends of
// a hit's window range are non-decreasing in g, so the number of sites that precede site (hit i, window w) in the
// reference's order (motif, window, position, strand) is
//     dst[i]                                   all windows of all earlier hits (exclusive prefix sum of the counts)
//   + (w - lo_i)                               the hit's own earlier windows
//   - sum_{i' < i} max(0, hi_i' - w)           earlier hits' windows that come AFTER w
//   + sum_{i' > i} max(0, min(w, hi_i' + 1) - lo_i')   later hits' windows that come BEFORE w
// where only hits of the same motif within one window length of g contribute to the two sums (a few at most, except
// in low-complexity floods where the walk is bounded by window * 2 strands).  A site is the first of its
// (motif, window) iff the previous hit of the motif does not reach window w.
__global__ void __launch_bounds__(256) sweep_scatter_kernel(int64_t n, const int64_t *__restrict__ motif_off, int32_t P,
                                                            const int32_t *__restrict__ width, const int64_t *__restrict__ pos,
                                                            const double *__restrict__ score, const int8_t *__restrict__ strand,
                                                            const uint64_t *__restrict__ dst, int32_t window, int32_t stride,
                                                            int64_t n_windows, int64_t *__restrict__ seq_idx_out,
                                                            int64_t *__restrict__ pos_out, double *__restrict__ score_out,
                                                            int8_t *__restrict__ strand_out,
                                                            unsigned long long *__restrict__ region_counts) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    int32_t m = -1;
    int n_first = 0;
    if (live) {
        m = motif_of_hit(motif_off, P, i);
        const int W = width[m];
        const int64_t first = motif_off[m], last = motif_off[m + 1];
        const int64_t g = pos[i];
        int64_t lo, hi;
        sweep_window_range(g, W, window, stride, n_windows, lo, hi);
        if (hi >= lo) {
            const double sc = score[i];
            const int8_t sd = strand[i];
            const int64_t base = (int64_t) dst[i];
            int64_t prev_hi = -1;                                    // window range end of the previous hit of this motif
            if (i > first) { int64_t l2; sweep_window_range(pos[i - 1], W, window, stride, n_windows, l2, prev_hi); if (prev_hi < l2) prev_hi = -1; }
            for (int64_t w = lo; w <= hi; w++) {
                int64_t idx = base + (w - lo);
                for (int64_t j = i - 1; j >= first; j--) {           // earlier hits still reaching past w
                    int64_t l2, h2;
                    sweep_window_range(pos[j], W, window, stride, n_windows, l2, h2);
                    if (h2 <= lo) break;                             // monotone: nothing further back reaches past lo <= w
                    if (h2 >= l2 && h2 > w) idx -= h2 - w;
                }
                for (int64_t j = i + 1; j < last; j++) {             // later hits that already started before w
                    int64_t l2, h2;
                    sweep_window_range(pos[j], W, window, stride, n_windows, l2, h2);
                    if (l2 >= hi) break;                             // monotone: nothing further on starts before hi >= w
                    if (h2 >= l2 && l2 < w) idx += (w < h2 + 1 ? w : h2 + 1) - l2;
                }
                seq_idx_out[idx] = w;
                pos_out[idx] = g - w * stride;
                score_out[idx] = sc;
                strand_out[idx] = sd;
                if (prev_hi < w) n_first++;
            }
        }
    }
    // windows with >= 1 site per motif (stats.py:29-31): one atomic per (wave, motif)
    unsigned long long todo = __ballot(live && n_first > 0);
    while (todo) {
        const int leader = __ffsll((long long) todo) - 1;
        const int32_t mm = __shfl(m, leader);
        const unsigned long long same = __ballot(live && n_first > 0 && m == mm);
        int v = (live && m == mm) ? n_first : 0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if ((int) (threadIdx.x & 63) == leader) atomicAdd(&region_counts[mm], (unsigned long long) v);
        todo &= ~same;
    }
}

// The hand-out of a COUNTS-ONLY sweep (MS_STREAM_NO_HITS; round 5): what the enrichment statistics read of a sweep is, per motif, the number
// of windows that hold >= 1 site (stats.py:29-31) -- plus, here, the number of sites.  Within a motif the hits are ordered by span
// position and both ends of a hit's window range are non-decreasing, so hit i is the FIRST site of exactly the windows of its range that
// the previous hit of the motif does not reach: max(0, hi_i - max(lo_i, prev_hi + 1) + 1).  One read per hit, no site is written
// (the full hand-out writes 25 bytes per (site, window): 36 GB per pass of a 3 Gbp genome).
__global__ void __launch_bounds__(256) sweep_countonly_kernel(int64_t n, const int64_t *__restrict__ motif_off, int32_t P,
                                                              const int32_t *__restrict__ width, const int64_t *__restrict__ pos,
                                                              int32_t window, int32_t stride, int64_t n_windows,
                                                              unsigned long long *__restrict__ region_counts, unsigned long long *__restrict__ n_sites /* [P] */) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    int32_t m = -1;
    int n_first = 0;
    unsigned long long mine = 0;
    if (live) {
        m = motif_of_hit(motif_off, P, i);
        const int W = width[m];
        int64_t lo, hi;
        sweep_window_range(pos[i], W, window, stride, n_windows, lo, hi);
        if (hi >= lo) {
            mine = (unsigned long long) (hi - lo + 1);
            int64_t prev_hi = -1;
            if (i > motif_off[m]) { int64_t l2; sweep_window_range(pos[i - 1], W, window, stride, n_windows, l2, prev_hi); if (prev_hi < l2) prev_hi = -1; }
            const int64_t from = lo > prev_hi + 1 ? lo : prev_hi + 1;
            n_first = hi >= from ? (int) (hi - from + 1) : 0;
        }
    }
    // per motif: windows with >= 1 site and sites -- one pair of atomics per (wave, motif), spread over the motifs' words (ONE counter for
    // the sites was tried first: 234 k atomics per span on one address, 2.6 ms per span at the ~90 M/s a single word sustains)
    unsigned long long todo = __ballot(live && mine > 0);
    while (todo) {
        const int leader = __ffsll((long long) todo) - 1;
        const int32_t mm = __shfl(m, leader);
        const unsigned long long same = __ballot(live && mine > 0 && m == mm);
        int v = (live && m == mm) ? n_first : 0;
        unsigned long long s64 = (live && m == mm) ? mine : 0ULL;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { v += __shfl_xor(v, o); s64 += __shfl_xor(s64, o); }
        if ((int) (threadIdx.x & 63) == leader) {
            if (v) atomicAdd(&region_counts[mm], (unsigned long long) v);
            atomicAdd(&n_sites[mm], s64);
        }
        todo &= ~same;
    }
}

// per-motif offsets of the handed-out sites: where the motif's first hit went
__global__ void sweep_offsets_kernel(const int64_t *__restrict__ motif_off, int32_t P, int64_t n, const uint64_t *__restrict__ dst,
                                     int64_t total, int64_t *__restrict__ out) {
    const int32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m > P) return;
    const int64_t f = m < P ? motif_off[m] : n;
    out[m] = f < n ? (int64_t) dst[f] : total;
}

__global__ void fill_nan_kernel(double *__restrict__ a, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = __longlong_as_double(0x7FF8000000000000LL);
}

// -------------------------------------------------------------------------- score --

// c_score (cscore.c:191-224): one thread per (sequence, motif); first W bases only.
__global__ void __launch_bounds__(256) score_kernel(const DevSeq S, const DevPwm Pw, int strand_mask,
                                                    double *__restrict__ out) {
    const int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t p = blockIdx.y;
    if (r >= S.R) return;
    const int W = Pw.width[p];
    const int64_t start = S.offsets[r];
    const int64_t len = S.offsets[r + 1] - start;
    const double2 *__restrict__ tab = Pw.tab2 + Pw.tab_off[p];
    double fwd = 0.0, rev = 0.0;
    const int n = (int) (len < W ? len : W);          // bases past the sequence end add nothing
    for (int c0 = 0; c0 < n; c0 += 32) {
        const uint64_t cw = code_window(S.codes, start + c0);
        const uint32_t nw = n_window(S.nmask, start + c0);
        const int m = (n - c0) < 32 ? (n - c0) : 32;
        for (int c = 0; c < m; c++) {
            if ((nw >> c) & 1u) continue;
            const uint32_t b = (uint32_t) (cw >> (2 * c)) & 3u;
            const double2 t = tab[(c0 + c) * 4 + b];
            fwd += t.x;
            rev += t.y;
        }
    }
    double s = 0.0;
    switch (strand_mask) {                               // cscore.c:208-222
        case 1: s = fwd; break;
        case 2: s = rev; break;
        case 3: s = fwd > rev ? fwd : rev; break;
    }
    out[(int64_t) p * S.R + r] = s / Pw.max_raw[p];
}

// out[k] = sorted[ranks[k]]  (ranks beyond the row give NaN)
__global__ void gather_ranks_kernel(const double *__restrict__ sorted, int64_t n, const int64_t *__restrict__ ranks,
                                    int32_t n_ranks, double *__restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_ranks) return;
    const int64_t r = ranks[k];
    out[k] = (r >= 0 && r < n) ? sorted[r] : __longlong_as_double(0x7FF8000000000000LL);
}

// --------------------------------------------------------------- on-device extraction --

// Regions cut out of a resident packed genome (replaces Scanner._extract_seq -> Genome.fetch_sequence
// -> pysam fetch, scanner.py:71-87 / genome/__init__.py:117-135): one thread per 32 output bases,
// which may straddle several regions.  src_start[r] is the region's first base in the genome's
// packed coordinates; dst_off[r] its first base in the output.
__global__ void __launch_bounds__(256) extract_kernel(const uint32_t *__restrict__ gcodes, const uint32_t *__restrict__ gnmask,
                                                      const int64_t *__restrict__ src_start, const int64_t *__restrict__ dst_off,
                                                      int64_t R, int64_t n_out, uint32_t *__restrict__ codes,
                                                      uint32_t *__restrict__ nmask) {
    const int64_t u = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t pos = u * 32;
    if (pos >= n_out) return;
    int64_t r = find_region_bsearch(dst_off, R, pos);
    uint64_t cw = 0;
    uint32_t nw = 0;
    int filled = 0;
    while (filled < 32 && pos + filled < n_out) {
        const int64_t d = pos + filled;
        while (dst_off[r + 1] <= d) r++;                              // skip empty regions
        const int64_t left = dst_off[r + 1] - d;
        const int seg = left < (int64_t) (32 - filled) ? (int) left : 32 - filled;
        const int64_t sp = src_start[r] + (d - dst_off[r]);
        const uint64_t scw = code_window(gcodes, sp);
        const uint32_t snw = n_window(gnmask, sp);
        const uint64_t m = seg >= 32 ? ~0ULL : ((1ULL << (2 * seg)) - 1ULL);
        cw |= (scw & m) << (2 * filled);
        nw |= (snw & low_mask(seg)) << filled;
        filled += seg;
    }
    codes[2 * u] = (uint32_t) cw;
    codes[2 * u + 1] = (uint32_t) (cw >> 32);
    nmask[u] = nw;
}

// ------------------------------------------------------------------- region hints --

// blk2reg[b] = region that holds position 64*b (part of the extraction stage, next to pack_kernel); blkinfo[b] = the same region with
// its own and the next two regions' starts RELATIVE to 64*b as 32-bit numbers -- everything rescore_kernel needs to place a position,
// in one 16-byte read (the fp64 stage pays per vector-memory instruction); region -1: a start lies more than 2^31 bases away, look it up
__global__ void __launch_bounds__(256) blk2reg_kernel(const int64_t *__restrict__ offsets, int64_t R, int64_t n_blocks,
                                                      int32_t *__restrict__ blk2reg, int4 *__restrict__ blkinfo, int all_far) {
    const int64_t b = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const int64_t r = find_region_bsearch(offsets, R, b * 64);
    blk2reg[b] = (int32_t) r;
    const int64_t base = b * 64;
    const int64_t o0 = offsets[r] - base, o1 = offsets[r + 1 <= R ? r + 1 : R] - base, o2 = offsets[r + 2 <= R ? r + 2 : (r + 1 <= R ? r + 1 : R)] - base;
    const bool fits = o0 > -(1LL << 31) && o1 < (1LL << 31) && o2 < (1LL << 31) && o1 > -(1LL << 31) && o2 > -(1LL << 31) && r < (1LL << 31);
    blkinfo[b] = fits && !all_far ? make_int4((int) r, (int) o0, (int) o1, (int) o2) : make_int4(-1, 0, 0, 0);     // (all_far: a test aid, MS_BLKINFO_FAR)
}

// ---------------------------------------------------------------- compact copy-out --

// coord = seq_idx << 32 | pos << 1 | (strand - 1): 8 bytes per hit on the host link instead of 17 (ms_result_hits_packed_host).
// shift > 0: the 4-byte form, coord32 = seq_idx << shift | pos << 1 | (strand - 1) (ms_result_hits_packed12_host: the caller has
// checked that every region index and position of the set fits).  bad[0] is set if a hit does not fit the format.
__global__ void __launch_bounds__(256) pack_hits_kernel(int64_t n, const unsigned long long *__restrict__ n_dev, const int64_t *__restrict__ seq_idx,
                                                        const int64_t *__restrict__ pos, const int8_t *__restrict__ strand,
                                                        uint64_t *__restrict__ coord, unsigned int *__restrict__ bad, int shift) {
    if (n_dev) { const unsigned long long nd = *n_dev; if ((unsigned long long) n > nd) n = (int64_t) nd; }     // (finalize_kernel: a predicted-size scan)
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t sq = (uint64_t) seq_idx[i], ps = (uint64_t) pos[i];
    if (shift > 0) {
        if ((ps >> (shift - 1)) != 0 || (sq >> (32 - shift)) != 0) *bad = 1u;
        reinterpret_cast<uint32_t *>(coord)[i] = (uint32_t) ((sq << shift) | (ps << 1) | (uint64_t) (strand[i] == 2 ? 1 : 0));
        return;
    }
    if ((sq >> 32) != 0 || (ps >> 31) != 0) *bad = 1u;
    coord[i] = (sq << 32) | ((ps & 0x7FFFFFFFull) << 1) | (uint64_t) (strand[i] == 2 ? 1 : 0);
}

int launch_pack_hits(int64_t n, const unsigned long long *n_dev, const int64_t *seq_idx, const int64_t *pos, const int8_t *strand, uint64_t *coord,
                     unsigned int *bad, hipStream_t st, int shift) {
    if (n == 0) return MS_OK;
    hipLaunchKernelGGL(pack_hits_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, n, n_dev, seq_idx, pos, strand, coord, bad, shift);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

// ---------------------------------------------------------------------- launchers --

int launch_extract(const uint32_t *gcodes, const uint32_t *gnmask, const int64_t *src_start, const int64_t *dst_off,
                   int64_t R, int64_t n_out, uint32_t *codes, uint32_t *nmask, hipStream_t st) {
    const int64_t n_units = (n_out + 31) / 32;
    if (n_units == 0) return MS_OK;
    hipLaunchKernelGGL(extract_kernel, dim3((unsigned) ((n_units + 255) / 256)), dim3(256), 0, st, gcodes, gnmask, src_start,
                       dst_off, R, n_out, codes, nmask);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_blk2reg(const int64_t *offsets, int64_t R, int64_t n_bases, int32_t *blk2reg, int4 *blkinfo, hipStream_t st) {
    const int64_t n_blocks = (n_bases + 63) / 64 + 1;
    const int all_far = measure_env("MS_BLKINFO_FAR") ? 1 : 0;          // test aid: every block record says "look the region up" (starts beyond 32 bits)
    hipLaunchKernelGGL(blk2reg_kernel, dim3((unsigned) ((n_blocks + 255) / 256)), dim3(256), 0, st, offsets, R, n_blocks,
                       blk2reg, blkinfo, all_far);
    MS_HIP(hipGetLastError());
    return MS_OK;
}


int launch_pack(const uint8_t *ascii, int64_t n_bases, uint32_t *codes, uint32_t *nmask, hipStream_t st) {
    const int64_t n_units = (n_bases + 31) / 32;
    if (n_units == 0) return MS_OK;
    const int aligned16 = (reinterpret_cast<uintptr_t>(ascii) & 15u) == 0;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned) ((n_units + 255) / 256)), dim3(256), 0, st, ascii, n_bases,
                       codes, nmask, n_units, aligned16);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

typedef void (*PfKernel)(const PfArgs);
static PfKernel pf_kernel(bool wide, bool meas, bool dense, int floor_ = 0) {                  // (dense: only without wide classes and outside the measurement instantiation)
    if (wide) return meas ? prefilter_f6_kernel<4, true> : prefilter_f6_kernel<4, false>;
    if (floor_ == 1) return prefilter_f6_kernel<2, false, false, 1>;
    if (floor_ == 2) return prefilter_f6_kernel<2, false, false, 2>;
    if (floor_ == 3) return prefilter_f6_kernel<2, false, false, 3>;
    if (floor_ == 4) return prefilter_f6_kernel<2, false, false, 4>;
    if (dense && !meas) return prefilter_f6_kernel<2, false, true>;
    return meas ? prefilter_f6_kernel<2, true> : prefilter_f6_kernel<2, false>;
}

int prefilter_set_lds(bool wide, bool meas, bool dense, size_t bytes, int floor_) {
    MS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pf_kernel(wide, meas, dense, floor_)), hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
    return MS_OK;
}

// wide: the plan holds row tiles of 3 or 4 k-blocks; dense: the form that decodes candidates in place (many hits per row tile)
int launch_prefilter(const PfArgs &A, bool wide, bool meas, bool dense, int blocks_per_tile, int n_tiles, size_t lds_bytes, hipStream_t st, int floor_) {
    const int64_t n_chunks = (A.n_bases + kPfThreads - 1) / kPfThreads;
    if (blocks_per_tile > n_chunks) blocks_per_tile = (int) n_chunks;
    hipLaunchKernelGGL(pf_kernel(wide, meas, dense, floor_), dim3((unsigned) blocks_per_tile, (unsigned) n_tiles), dim3(kPfThreads), lds_bytes, st, A);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_exact_all(const DevSeq &S, const DevPwm &Pw, const int32_t *motifs, int32_t n_motifs, int strand_mask,
                     const HitOut &H, hipStream_t st, int max_width) {
    if (S.n_bases == 0 || n_motifs == 0) return MS_OK;
    const bool tiled = max_width >= 1 && max_width <= kExactTileMaxW && !measure_env("MS_EXACT_UNTILED");      // (A/B and test aid: the round-1 kernel)
    const size_t lds = tiled ? ((size_t) max_width * 4 + 1) * sizeof(double2) : 0;
    if (tiled && lds > 48 * 1024)                                    // (motifs of more than 767 columns: rare enough to ask the driver every time, on whatever device is current)
        MS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(exact_tiled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    for (int32_t m0 = 0; m0 < n_motifs; m0 += 32768) {               // grid.y limit
        const int32_t n = n_motifs - m0 < 32768 ? n_motifs - m0 : 32768;
        if (tiled) {
            dim3 grid((unsigned) ((S.n_bases + 256 * kExactIter - 1) / (256 * kExactIter)), (unsigned) n);
            hipLaunchKernelGGL(exact_tiled_kernel, grid, dim3(256), lds, st, S, Pw, motifs + m0, strand_mask, H);
        } else {
            dim3 grid((unsigned) ((S.n_bases + 255) / 256), (unsigned) n);
            hipLaunchKernelGGL(exact_all_kernel, grid, dim3(256), 0, st, S, Pw, motifs + m0, strand_mask, H);
        }
        MS_HIP(hipGetLastError());
    }
    return MS_OK;
}

// ---- the fp64 stage for LONG candidate lists: chunks of the list in motif order, the window carried through the sort.
// A scattered read costs the texture addressers ~1.9 cycles per distinct 64-byte line its 64 lanes touch, whatever its width
// (tools/ubench/gather_rate.hip: 122 cycles for 64 lines, 22 for 8), and ~16 of the ~21 reads per candidate are table entries -- one
// line per (motif, column).  In list order a wave's lanes hold ~16 different motifs; in motif order far fewer -- but then the three
// position-bound reads (block record, code words, mask words), whose lanes sit in one or two pre-filter units in list order, would
// touch 64 lines each.  So a block takes 4096 candidates, reads the position-bound data in LIST order, packs what the scoring needs
// into 24 bytes per candidate -- the window's 32 codes, its mask bits, (group, flags), the hit coordinate and how much room the region
// leaves -- brings THAT into (table group, first flagged field) order with a counting sort in LDS on a hash of the record (no memory
// read), and the scoring pass reads only per-motif data.  profiles/r03z_rescore_sorted.log: 2.01 ms per 500 Mbase against 2.73 in list
// order (rescore_kernel, which short lists keep: a chunk per block leaves most of the device idle below ~2e6 candidates).
constexpr int kRwThreads = 1024;
constexpr int kRwPerThread = 4;
constexpr int kRwChunk = kRwThreads * kRwPerThread;       // 4096 candidates: 96 KB of LDS
constexpr int kRwBins = 4096;
typedef HitStageN<2048> RwStage;

__global__ void __launch_bounds__(kRwThreads) rescore_carry_kernel(const DevSeq S, const DevPwm Pw, const uint64_t *__restrict__ cand,
                                                                   const unsigned long long *__restrict__ n_cand, uint64_t n_static, uint64_t cand_cap,
                                                                   const FieldMeta *__restrict__ field_meta, int strand_mask, const HitOut H) {
    extern __shared__ uint4 rw_lds4[];
    uint64_t *s_cw = reinterpret_cast<uint64_t *>(rw_lds4);                                 // [kRwChunk] the window's codes
    uint64_t *s_gk = s_cw + kRwChunk;                                                       // [kRwChunk] hit coordinate << 8 | room (bases to the region's end, <= 255)
    uint32_t *s_nw = reinterpret_cast<uint32_t *>(s_gk + kRwChunk);                         // [kRwChunk] the window's non-ACGT bits
    uint32_t *s_gf = s_nw + kRwChunk;                                                       // [kRwChunk] group << 16 | flags (0: nothing)
    uint32_t *bins = s_gf + kRwChunk;                                                       // [kRwBins]
    uint32_t *wave_tot = bins + kRwBins;                                                    // [16]
    RwStage &st = *reinterpret_cast<RwStage *>(wave_tot + 32);
    if (threadIdx.x == 0) st.n = 0;
    unsigned long long n = n_static + *n_cand;
    if (n > cand_cap) n = cand_cap;
    const bool both = strand_mask == 3;
    const int4 *__restrict__ meta4 = reinterpret_cast<const int4 *>(field_meta);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t pmask = H.pbits ? ((1ULL << H.pbits) - 1ULL) : 0ULL;
    const unsigned long long n_chunks = (n + kRwChunk - 1) / kRwChunk;
    for (unsigned long long ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        for (uint32_t i = threadIdx.x; i < (uint32_t) kRwBins; i += kRwThreads) bins[i] = 0;
        __syncthreads();
        {
            // ---- list order: everything that depends on the POSITION
            uint64_t cw[kRwPerThread], gk[kRwPerThread];
            uint32_t nw[kRwPerThread], gf[kRwPerThread], key[kRwPerThread], rank[kRwPerThread];
            int64_t g[kRwPerThread];
            int4 bi[kRwPerThread];
#pragma unroll
            for (int k = 0; k < kRwPerThread; k++) {
                const unsigned long long i = ch * kRwChunk + (unsigned long long) k * kRwThreads + threadIdx.x;
                const uint64_t rec = i < n ? cand[i] : 0ULL;
                g[k] = (int64_t) (rec >> 30);
                gf[k] = (uint32_t) rec & 0x3FFFFFFFu;                                       // group << 16 | flags
                if (!(gf[k] & 0xFFFFu)) { gf[k] = 0; g[k] = 0; }
                bi[k] = S.blkinfo[g[k] >> 6];
                cw[k] = code_window(S.codes, g[k]);
                nw[k] = n_window(S.nmask, g[k]);
            }
#pragma unroll
            for (int k = 0; k < kRwPerThread; k++) {
                const int64_t base = g[k] & ~(int64_t) 63;
                int64_t lo = bi[k].x, o0 = base + bi[k].y, o1 = base + bi[k].z, o2 = base + bi[k].w, r, beg, end;
                if (bi[k].x < 0) {
                    lo = S.blk2reg[g[k] >> 6];
                    o0 = S.offsets[lo]; o1 = S.offsets[lo + 1];
                    o2 = lo + 2 <= S.R ? S.offsets[lo + 2] : o1;
                }
                if (g[k] < o1) { r = lo; beg = o0; end = o1; }
                else if (g[k] < o2) { r = lo + 1; beg = o1; end = o2; }
                else { r = find_region(S, g[k]); beg = S.offsets[r]; end = S.offsets[r + 1]; }      // tiny regions
                const uint64_t coord = H.pbits ? (((uint64_t) r << H.pbits) | (uint64_t) (g[k] - beg)) : (uint64_t) g[k];
                const int64_t room = end - g[k];
                gk[k] = (coord << 8) | (uint64_t) (room > 255 ? 255 : (room < 0 ? 0 : room));
                const uint32_t f = gf[k] & 0xFFFFu;
                const uint32_t fl = both ? (f | (f >> 1)) & 0x5555u : f;
                key[k] = fl ? (((gf[k] >> 16) << 4) | (uint32_t) (__ffs((int) fl) - 1)) & (uint32_t) (kRwBins - 1) : (uint32_t) (kRwBins - 1);
            }
#pragma unroll
            for (int k = 0; k < kRwPerThread; k++) rank[k] = atomicAdd(&bins[key[k]], 1u);
            __syncthreads();
            {   // exclusive prefix over the bins: four per thread, wave scans, wave totals
                const uint32_t b0 = bins[4 * threadIdx.x], b1 = bins[4 * threadIdx.x + 1], b2 = bins[4 * threadIdx.x + 2], b3 = bins[4 * threadIdx.x + 3];
                const uint32_t tot = b0 + b1 + b2 + b3;
                uint32_t v = tot;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = __shfl_up(v, d);
                    if ((int) lane >= d) v += o;
                }
                if (lane == 63u) wave_tot[wave] = v;
                __syncthreads();
                uint32_t base = 0;
                for (uint32_t w = 0; w < wave; w++) base += wave_tot[w];
                const uint32_t excl = base + v - tot;
                bins[4 * threadIdx.x] = excl;
                bins[4 * threadIdx.x + 1] = excl + b0;
                bins[4 * threadIdx.x + 2] = excl + b0 + b1;
                bins[4 * threadIdx.x + 3] = excl + b0 + b1 + b2;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kRwPerThread; k++) {
                const uint32_t at = bins[key[k]] + rank[k];
                s_cw[at] = cw[k]; s_gk[at] = gk[k]; s_nw[at] = nw[k]; s_gf[at] = gf[k];
            }
        }
        __syncthreads();
        // ---- motif order: two rounds of two candidates per thread, per-motif reads only
        for (int round = 0; round < 2; round++) {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint32_t at = (uint32_t) (round * 2 + u) * kRwThreads + threadIdx.x;
                const uint32_t gfu = s_gf[at];
                const uint32_t f = gfu & 0xFFFFu;
                uint32_t flags = both ? (f | (f >> 1)) & 0x5555u : f;
                if (!flags) continue;
                const uint64_t cwu = s_cw[at], gku = s_gk[at];
                const uint32_t nwu = s_nw[at];
                const int32_t group = (int32_t) (gfu >> 16);
                const int room = (int) (gku & 0xFFu);
                const int64_t coord = (int64_t) (gku >> 8);
                while (flags) {
                    const int field = __ffs((int) flags) - 1;
                    flags &= flags - 1u;
                    const int4 f4 = meta4[group * kGroupFields + field];
                    const int32_t m = f4.x;
                    const int w = f4.y;
                    if (m < 0) continue;
                    if (w > room) continue;                                  // window runs past its region (cscore.c:340)
                    double fwd, rev;
                    if (w <= 32 && Pw.tab32) score_window32(Pw.tab2, (uint32_t) f4.z, Pw.zero_bytes, w, cwu, nwu, fwd, rev);
                    else {
                        const int64_t gpos = H.pbits ? S.offsets[coord >> H.pbits] + (int64_t) ((uint64_t) coord & pmask) : coord;
                        score_window(S, Pw.tab2 + Pw.tab_off[m], w, gpos, fwd, rev);
                    }
                    test_and_stage(st, H, Pw, (uint32_t) m, coord, fwd, rev, strand_mask, __int_as_float(f4.w));
                }
            }
            stage_flush(st, H);
        }
    }
}

size_t rescore_carry_lds_bytes() { return (size_t) kRwChunk * 24 + (size_t) kRwBins * 4 + 32 * 4 + sizeof(RwStage) + 64; }
int rescore_carry_set_lds() {
    MS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(rescore_carry_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) rescore_carry_lds_bytes()));
    return MS_OK;
}
int launch_rescore_carry(const DevSeq &S, const DevPwm &Pw, const uint64_t *cand, const unsigned long long *n_cand, uint64_t n_static,
                         uint64_t cand_cap, const FieldMeta *field_meta, int strand_mask, const HitOut &H, int n_blocks, hipStream_t st) {
    hipLaunchKernelGGL(rescore_carry_kernel, dim3((unsigned) n_blocks), dim3(kRwThreads), rescore_carry_lds_bytes(), st, S, Pw, cand, n_cand, n_static,
                       cand_cap, field_meta, strand_mask, H);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_rescore(const DevSeq &S, const DevPwm &Pw, const uint64_t *cand, const unsigned long long *n_cand, uint64_t n_static,
                   uint64_t cand_cap, const FieldMeta *field_meta, int strand_mask, const HitOut &H, int n_blocks,
                   hipStream_t st) {
    hipLaunchKernelGGL(rescore_kernel, dim3((unsigned) n_blocks), dim3(256), 0, st, S, Pw, cand, n_cand, n_static, cand_cap,
                       field_meta, strand_mask, H);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_finalize(const uint64_t *keys, int64_t n, const unsigned long long *n_dev, int gbits, int rbits, int pbits, int32_t P, const DevSeq &S,
                    int64_t *seq_idx, int64_t *pos, int8_t *strand, int64_t *motif_first, unsigned long long *region_counts,
                    hipStream_t st) {
    if (n == 0 || n_dev) {                           // no hits: every per-motif offset is 0 (with n_dev the kernel overwrites them unless the count is 0)
        MS_HIP(hipMemsetAsync(motif_first, 0, ((size_t) P + 1) * sizeof(int64_t), st));
        if (n == 0) return MS_OK;
    }
    if (pbits > 0) {
        hipLaunchKernelGGL(finalize_rp_kernel, dim3((unsigned) ((n + 1023) / 1024)), dim3(256), 0, st, keys, n, n_dev, rbits, pbits, P,
                           seq_idx, pos, strand, motif_first, region_counts);
        MS_HIP(hipGetLastError());
        return MS_OK;
    }
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, keys, n, n_dev, gbits, P, S,
                       seq_idx, pos, strand, motif_first, region_counts);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

// keys[i] = all ones for i in [min(*n_dev, cap), cap): the unused rest of a predicted-size hit list sorts behind every hit
__global__ void __launch_bounds__(256) fill_tail_kernel(uint64_t *__restrict__ keys, const unsigned long long *__restrict__ n_dev, uint64_t cap) {
    const unsigned long long n = *n_dev < cap ? *n_dev : cap;
    for (unsigned long long i = n + (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (unsigned long long) gridDim.x * blockDim.x)
        keys[i] = ~0ULL;
}

// The radix sort orders a scan's hits over the key bits ABOVE kSortLowBits only (one eight-bit pass fewer over 62 M pairs of 16 bytes);
// hits that agree in those bits -- the same motif, region and 128-base stretch: a motif's two strands at one position, mostly -- are
// neighbours afterwards, in the order the list held them.  This kernel finishes the order: the first hit of every such run sorts its
// run in place by the whole key (runs hold <= 2^kSortLowBits hits: the keys of a scan are distinct; all-ones padding keys are left alone).
__global__ void __launch_bounds__(256) sort_fixup_kernel(uint64_t *__restrict__ keys, double *__restrict__ vals, int64_t n, const unsigned long long *__restrict__ n_dev) {
    if (n_dev) { const unsigned long long nd = *n_dev; if ((unsigned long long) n > nd) n = (int64_t) nd; }
    const int64_t i0 = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) * 4;     // four consecutive hits per thread: two 16-byte reads
    if (i0 >= n) return;
    uint64_t k[4] = {0, 0, 0, 0};
    uint64_t hi[6];                                                               // the high bits of hits i0 - 1 ... i0 + 4 (all-ones: none)
    hi[0] = i0 > 0 ? keys[i0 - 1] >> kSortLowBits : ~0ULL;
    if (i0 + 4 <= n) {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(keys + i0), b = *reinterpret_cast<const ulonglong2 *>(keys + i0 + 2);
        k[0] = a.x; k[1] = a.y; k[2] = b.x; k[3] = b.y;
    } else {
        for (int q = 0; q < 4; q++) k[q] = i0 + q < n ? keys[i0 + q] : ~0ULL;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) hi[1 + q] = i0 + q < n ? k[q] >> kSortLowBits : ~0ULL;
    hi[5] = i0 + 4 < n ? keys[i0 + 4] >> kSortLowBits : ~0ULL;
    // (the high bits of a run's members do not change while another thread sorts the run: what is compared here is stable)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int64_t i = i0 + q;
        if (i + 1 >= n || hi[1 + q] == hi[q] || hi[1 + q] != hi[2 + q]) continue;   // not the first of a run of two or more
        if (q < 3 && hi[3 + q] != hi[1 + q]) {
            // a run of exactly two, both in this thread's registers (a motif's two strands at one position: nearly every run):
            // half of them are in order already and touch nothing more
            if (k[q] > k[q + 1]) {
                keys[i] = k[q + 1];
                keys[i + 1] = k[q];
                const double v0 = vals[i], v1 = vals[i + 1];
                vals[i] = v1;
                vals[i + 1] = v0;
            }
            continue;
        }
        const uint64_t h = hi[1 + q];
        int64_t len = 2;
        while (i + len < n && len < ((int64_t) 1 << kSortLowBits) && (keys[i + len] >> kSortLowBits) == h) len++;
        for (int64_t a = 1; a < len; a++) {                                      // insertion sort: short runs
            const uint64_t ka = keys[i + a];
            const double va = vals[i + a];
            int64_t b = a;
            while (b > 0 && keys[i + b - 1] > ka) { keys[i + b] = keys[i + b - 1]; vals[i + b] = vals[i + b - 1]; b--; }
            keys[i + b] = ka;
            vals[i + b] = va;
        }
    }
}

int launch_sort_fixup(uint64_t *keys, double *vals, int64_t n, const unsigned long long *n_dev, hipStream_t st) {
    if (n == 0) return MS_OK;
    hipLaunchKernelGGL(sort_fixup_kernel, dim3((unsigned) ((n + 1023) / 1024)), dim3(256), 0, st, keys, vals, n, n_dev);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_fill_tail(uint64_t *keys, const unsigned long long *n_dev, uint64_t cap, hipStream_t st) {
    if (cap == 0) return MS_OK;
    hipLaunchKernelGGL(fill_tail_kernel, dim3(256), dim3(256), 0, st, keys, n_dev, cap);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_sweep_count(int64_t n, const int64_t *motif_off, int32_t P, const int32_t *width, const int64_t *pos,
                       int32_t window, int32_t stride, int64_t n_windows, uint32_t *cnt, hipStream_t st) {
    if (n == 0) return MS_OK;
    hipLaunchKernelGGL(sweep_count_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, n, motif_off, P, width, pos,
                       window, stride, n_windows, cnt);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_sweep_countonly(int64_t n, const int64_t *motif_off, int32_t P, const int32_t *width, const int64_t *pos,
                           int32_t window, int32_t stride, int64_t n_windows, unsigned long long *region_counts, unsigned long long *n_sites, hipStream_t st) {
    if (n == 0) return MS_OK;
    hipLaunchKernelGGL(sweep_countonly_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, n, motif_off, P, width, pos,
                       window, stride, n_windows, region_counts, n_sites);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_sweep_scatter(int64_t n, const int64_t *motif_off, int32_t P, const int32_t *width, const int64_t *pos,
                         const double *score, const int8_t *strand, const uint64_t *dst, int32_t window, int32_t stride,
                         int64_t n_windows, int64_t total, int64_t *seq_idx_out, int64_t *pos_out, double *score_out,
                         int8_t *strand_out, int64_t *motif_off_out, unsigned long long *region_counts, hipStream_t st) {
    hipLaunchKernelGGL(sweep_offsets_kernel, dim3((unsigned) ((P + 1 + 255) / 256)), dim3(256), 0, st, motif_off, P, n, dst, total,
                       motif_off_out);
    MS_HIP(hipGetLastError());
    if (n == 0) return MS_OK;
    hipLaunchKernelGGL(sweep_scatter_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, n, motif_off, P, width, pos,
                       score, strand, dst, window, stride, n_windows, seq_idx_out, pos_out, score_out, strand_out, region_counts);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_dedup(int64_t n, const int64_t *motif_off, int32_t P, const int32_t *width, const int64_t *seq_idx,
                 const int64_t *pos, const double *score, const int8_t *strand, uint32_t *keep, hipStream_t st) {
    if (n == 0) return MS_OK;
    hipLaunchKernelGGL(dedup_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, n, motif_off, P, width, seq_idx,
                       pos, score, strand, keep);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_compact_hits(int64_t n, const uint32_t *keep, const uint64_t *dst, const int64_t *seq_in, const int64_t *pos_in,
                        const double *score_in, const int8_t *strand_in, int64_t *seq_out, int64_t *pos_out,
                        double *score_out, int8_t *strand_out, const int64_t *off_in, int32_t P, int64_t *off_out,
                        hipStream_t st) {
    hipLaunchKernelGGL(remap_offsets_kernel, dim3((unsigned) ((P + 1 + 255) / 256)), dim3(256), 0, st, off_in, P, n, dst, keep, off_out);
    MS_HIP(hipGetLastError());
    if (n == 0) return MS_OK;
    hipLaunchKernelGGL(compact_hits_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, n, keep, dst, seq_in, pos_in,
                       score_in, strand_in, seq_out, pos_out, score_out, strand_out);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_site_tables(int64_t n, const int64_t *motif_off, int32_t P, int64_t R, const int64_t *seq_idx,
                       const double *score, int32_t *n_sites, double *max_score, hipStream_t st) {
    const int64_t cells = (int64_t) P * R;
    if (cells == 0) return MS_OK;
    MS_HIP(hipMemsetAsync(n_sites, 0, (size_t) cells * sizeof(int32_t), st));
    hipLaunchKernelGGL(fill_nan_kernel, dim3((unsigned) ((cells + 255) / 256)), dim3(256), 0, st, max_score, cells);
    MS_HIP(hipGetLastError());
    if (n == 0) return MS_OK;
    hipLaunchKernelGGL(site_tables_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, n, motif_off, P, R, seq_idx,
                       score, n_sites, max_score);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_gather_ranks(const double *sorted, int64_t n, const int64_t *ranks, int32_t n_ranks, double *out, hipStream_t st) {
    if (n_ranks <= 0) return MS_OK;
    hipLaunchKernelGGL(gather_ranks_kernel, dim3((unsigned) ((n_ranks + 63) / 64)), dim3(64), 0, st, sorted, n, ranks, n_ranks, out);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_score(const DevSeq &S, const DevPwm &Pw, int strand_mask, double *out, hipStream_t st) {
    if (S.R == 0 || Pw.P == 0) return MS_OK;
    for (int32_t p0 = 0; p0 < Pw.P; p0 += 32768) {
        const int32_t n = Pw.P - p0 < 32768 ? Pw.P - p0 : 32768;
        DevPwm sub = Pw;
        sub.tab_off += p0; sub.width += p0; sub.max_raw += p0; sub.cutoff += p0; sub.raw_floor += p0; sub.P = n;
        dim3 grid((unsigned) ((S.R + 255) / 256), (unsigned) n);
        hipLaunchKernelGGL(score_kernel, grid, dim3(256), 0, st, S, sub, strand_mask, out + (int64_t) p0 * S.R);
        MS_HIP(hipGetLastError());
    }
    return MS_OK;
}

}  // namespace ms
