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
//   prefilter_mfma_kernel  rigorous integer upper bound of both strand scores for EVERY window as an int8
//                     one-hot product on the matrix cores (v_mfma_i32_32x32x32_i8); emits candidates
//   prefilter_kernel  the same bound from packed 10/16-bit 2-mer fields read per lane from LDS
//                     (engine 0, MS_PF_ENGINE=0: A/B reference)
//   nlist/neval       fp64 scoring of the windows that overlap a non-ACGT base
//   exact_all_kernel  fp64 scoring of every window for motifs the pre-filter cannot take
//   rescore_kernel    fp64 scoring of the candidates, in the reference's order of operations,
//                     and the reference's hit test (cscore.c:356-358, 373-375)
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
    if (H.entries) {                                     // one entry per window: rescore_ordered_kernel repeats this very test
        if (hit_f || hit_r) {
            const unsigned long long i = atomicAdd(H.n_hits, 1ULL);
            if (i < H.cap) H.keys[i] = ((uint64_t) motif << H.gbits) | (uint64_t) g;
        }
        return;
    }
    if (hit_f) emit_hit(H, motif, g, 0u, s_f);
    if (hit_r) emit_hit(H, motif, g, 1u, s_r);
}

// Block-level staging of hits in LDS: one global atomicAdd per ~2000 hits instead of one per wave
// (all hit writers of the chip share ONE counter word; it sustains only ~90 M atomics/s).
constexpr int kHitStage = 2048;
struct HitStage {
    uint64_t keys[kHitStage];
    double vals[kHitStage];
    unsigned int n;
    unsigned long long base;
};

__device__ __forceinline__ void stage_hit(HitStage &st, const HitOut &H, uint32_t motif, int64_t g, uint32_t sbit, double score) {
    const unsigned int i = atomicAdd(&st.n, 1u);
    if (i < (unsigned int) kHitStage) {
        st.keys[i] = ((uint64_t) motif << (H.gbits + 1)) | ((uint64_t) g << 1) | sbit;
        st.vals[i] = score;
    } else {
        emit_hit(H, motif, g, sbit, score);            // stage full: straight to HBM
    }
}

// all threads of the block, at a block-uniform point
__device__ __forceinline__ void stage_flush(HitStage &st, const HitOut &H) {
    __syncthreads();
    const unsigned int n = st.n < (unsigned int) kHitStage ? st.n : (unsigned int) kHitStage;
    if (threadIdx.x == 0 && n > 0) st.base = atomicAdd(H.n_hits, (unsigned long long) n);
    __syncthreads();
    const unsigned long long base = st.base;
    for (unsigned int i = threadIdx.x; i < n; i += blockDim.x)
        if (base + i < H.cap) { H.keys[base + i] = st.keys[i]; H.vals[base + i] = st.vals[i]; }
    __syncthreads();
    if (threadIdx.x == 0) st.n = 0;
    __syncthreads();
}

__device__ __forceinline__ void test_and_stage(HitStage &st, const HitOut &H, const DevPwm &Pw, uint32_t motif, int64_t g,
                                               double fwd, double rev, int strand_mask) {
    const double floor_ = Pw.raw_floor[motif];               // see test_and_emit
    const bool try_f = (strand_mask & 1) && !(fwd < floor_);
    const bool try_r = (strand_mask & 2) && !(rev < floor_);
    if (!try_f && !try_r) return;
    const double max_raw = Pw.max_raw[motif];
    const double cutoff = Pw.cutoff[motif];
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

__device__ __forceinline__ void add4(uint4 &a, const uint4 &b) {
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
}

// ---- candidate hand-off --------------------------------------------------------------------
// Candidates are ~2e-4 of the (window, motif) pairs, i.e. about one wave in ten finds one in a
// pair of table groups.  One global atomic per find would put every wave of the chip on ONE address
// (measured: the whole kernel then runs at the ~90 M atomics/s a single word sustains).  So each
// wave appends to its own queue in LDS with ballot/mbcnt ranks (no atomics at all) and spills it
// to the global list with a single atomicAdd per >= 64 entries.  A record is per LANE and per
// table group: position, group, and one flag bit per field (motif slot, strand) -- rescore_kernel
// expands the flags.

__device__ __noinline__ void wq_flush(uint64_t *__restrict__ wbuf, uint32_t n, uint64_t *__restrict__ cand,
                                      unsigned long long *__restrict__ n_cand, uint64_t cand_cap) {
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(n_cand, (unsigned long long) n);
    base = __shfl(base, 0);
    for (uint32_t i = lane; i < n; i += 64)
        if (base + i < cand_cap) cand[base + i] = wbuf[i];
}

// Field geometry of a 32-bit table word at FB bits per field.
template <int FB> struct Fields {
    static constexpr int NF = 32 / FB;                                   // fields per word
    static constexpr uint32_t low() { uint32_t m = 0; for (int f = 0; f < NF; f++) m |= 1u << (f * FB); return m; }
    static constexpr uint32_t kLow = low();                              // bit 0 of every field
    static constexpr uint32_t kTop = kLow << (FB - 1);                   // top (flag) bit of every field
};

// 16-byte entry -> flags: bit n = field n reached its top bit (field n: word n & 3, field n >> 2)
template <int FB>
__device__ __forceinline__ uint32_t group_flags(const uint4 &a) {
    uint32_t f = (a.x >> (FB - 1)) & Fields<FB>::kLow;
    f |= ((a.y >> (FB - 1)) & Fields<FB>::kLow) << 1;
    f |= ((a.z >> (FB - 1)) & Fields<FB>::kLow) << 2;
    f |= ((a.w >> (FB - 1)) & Fields<FB>::kLow) << 3;
    uint32_t out = f & 0xFu;
#pragma unroll
    for (int k = 1; k < Fields<FB>::NF; k++) out |= ((f >> (k * FB)) & 0xFu) << (4 * k);
    return out;
}

struct PfWave {
    uint64_t *wbuf;      // this wave's queue in LDS (kWqCap entries)
    uint32_t n;          // entries queued (wave-uniform)
    int64_t g;           // this lane's window start
    bool live;           // g < n_bases
};

// Entered by the WHOLE wave (uniform branch); appends one record per lane that flagged anything in
// this table group.
__device__ __forceinline__ void emit_flags(const PfArgs &A, PfWave &W, uint32_t flags, int32_t group) {
    const bool flagged = W.live && flags != 0;
    const unsigned long long mask = __ballot(flagged);
    if (mask == 0) return;
    const uint32_t n_new = (uint32_t) __popcll(mask);
    if (W.n + n_new > (uint32_t) kWqCap) {
        wq_flush(W.wbuf, W.n, A.cand, A.n_cand, A.cand_cap);
        W.n = 0;
    }
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
    if (flagged) W.wbuf[W.n + rank] = cand_pack((uint64_t) W.g, (uint32_t) group, flags);
    W.n += n_new;
}

// One loop trip: NG consecutive table groups of a class for the lane's window start.  All NG*G
// reads are issued as one batch (at most 8 rows at a time when that is more than 16 reads), then
// added, then ONE test decides whether any of the 64 lanes flagged any field of the NG groups.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ds_read_b128 issued by hand: hipcc's scheduler otherwise splits a trip's reads into per-group
// batches of G with a full lgkmcnt(0) drain between them (kernel .s), i.e. keeps only ~6 reads in
// flight per wave.  The loads below are issued back to back; ONE s_waitcnt lgkmcnt(0) +
// sched_barrier follows (hipcc tracks neither inline-asm loads nor their waits).  lgkmcnt(0) also
// covers any scalar load in flight, so no counted wait has to reason about SMEM's out-of-order
// returns.
template <int OFF>
__device__ __forceinline__ u32x4 lds_read128_asm(uint32_t byte_addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(byte_addr), "n"(OFF));
    return r;
}

// s_waitcnt lgkmcnt(rows_left * NG), as an immediate
template <int NG>
__device__ __forceinline__ void wait_lgkm(int rows_left) {
    switch (rows_left * NG) {
        case 0: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt lgkmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt lgkmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt lgkmcnt(14)" ::: "memory"); break;
        default: asm volatile("s_waitcnt lgkmcnt(14)" ::: "memory"); break;   // 15 cannot be relied on (4-bit counter): a stricter wait is never wrong
    }
}

template <int G, int NG, int FB, bool ASM, bool PROG>
__device__ __forceinline__ void prefilter_trip(const PfArgs &A, PfWave &W, const uint4 *__restrict__ lds4,
                                               const uint32_t (&a)[G], int32_t group) {
    uint4 acc[NG];
    constexpr int B = (NG * G <= 16) ? G : (16 / NG < 1 ? 1 : 16 / NG);      // rows per batch
    if constexpr (ASM) {
        // a[] holds BYTE addresses here
#pragma unroll
        for (int k0 = 0; k0 < G; k0 += B) {
            u32x4 r[NG][B];
#pragma unroll
            for (int k = 0; k < B; k++)
                if (k0 + k < G) {
                    if constexpr (NG >= 1) r[0][k] = lds_read128_asm<0>(a[k0 + k]);
                    if constexpr (NG >= 2) r[1][k] = lds_read128_asm<1 * G * 256>(a[k0 + k]);
                    if constexpr (NG >= 3) r[2][k] = lds_read128_asm<2 * G * 256>(a[k0 + k]);
                    if constexpr (NG >= 4) r[3][k] = lds_read128_asm<3 * G * 256>(a[k0 + k]);
                }
            if constexpr (!PROG) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < B; k++)
                if (k0 + k < G) {
                    if constexpr (PROG) {
                        // LDS returns in order and nothing else is on lgkmcnt inside the trip loop: row k of
                        // all NG groups is back once at most (rows still behind it) x NG reads are outstanding
                        const int rows_in_batch = (G - k0) < B ? (G - k0) : B;       // folds after unrolling
                        wait_lgkm<NG>(rows_in_batch - 1 - k);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int u = 0; u < NG; u++) {
                        if (k0 + k == 0) { acc[u].x = r[u][k].x; acc[u].y = r[u][k].y; acc[u].z = r[u][k].z; acc[u].w = r[u][k].w; }
                        else { acc[u].x += r[u][k].x; acc[u].y += r[u][k].y; acc[u].z += r[u][k].z; acc[u].w += r[u][k].w; }
                    }
                }
        }
    } else {
#pragma unroll
        for (int k0 = 0; k0 < G; k0 += B) {
            uint4 r[NG][B];
#pragma unroll
            for (int k = 0; k < B; k++)
                if (k0 + k < G) {
#pragma unroll
                    for (int u = 0; u < NG; u++) r[u][k] = lds4[a[k0 + k] + u * G * 16];
                }
#pragma unroll
            for (int k = 0; k < B; k++)
                if (k0 + k < G) {
#pragma unroll
                    for (int u = 0; u < NG; u++) {
                        if (k0 + k == 0) acc[u] = r[u][k];
                        else add4(acc[u], r[u][k]);
                    }
                }
        }
    }
    uint32_t any[NG], all = 0;
#pragma unroll
    for (int u = 0; u < NG; u++) {
        any[u] = (acc[u].x | acc[u].y | acc[u].z | acc[u].w) & Fields<FB>::kTop;
        all |= any[u];
    }
    if (__any(all != 0) && !A.no_emit) {
        // usually only one of the groups has a flagged lane: decode flags only for that one
#pragma unroll
        for (int u = 0; u < NG; u++)
            if (NG == 1 || __any(any[u] != 0)) emit_flags(A, W, group_flags<FB>(acc[u]), group + u);
    }
}

// All table groups of one class (same 2-mer count G, same field width FB): per group G LDS reads
// of 16 bytes (2 * 32/FB motifs x {fwd,rev} fields) and (G-1) x 4 packed adds.  A table row of
// one (group, 2-mer position) is 16 codes x 16 B = 256 B = every LDS bank exactly once, so the
// read is conflict-free whatever the codes are (SQ_LDS_BANK_CONFLICT = 0, profiles/).
//
// The kernel is bound by LDS bandwidth, so a trip covers as many groups as give 12-16 reads in
// flight per wave (4 groups of narrow motifs, 2 of wide ones); the few groups left over at the
// end of a class take one smaller trip.  Variants (A/B runs, tools/pf_variants.py): V = 0 two groups
// per trip; V = 1 trips sized by width, reads left to hipcc; V = 3 reads issued by hand, one wait;
// V = 4 (default) reads issued by hand, counted waits so the adds start as rows arrive.  (A fully
// double-buffered form -- next trip's reads queued behind the current one's -- measured 6 % slower:
// one group per trip costs more test instructions than the shorter LDS queue gaps win.)
template <int G, int V, int FB>
__device__ __forceinline__ void prefilter_class(const PfArgs &A, const uint4 *__restrict__ lds4, uint32_t base16,
                                                int n_groups, int32_t first_group, const uint64_t cw, PfWave &W) {
    constexpr bool ASM = V == 3 || V == 4;
    constexpr bool PROG = V == 4;
    constexpr uint32_t unit = ASM ? 16u : 1u;          // a[] in bytes (hand-issued reads) or in 16-byte entries
    uint32_t a[G];
#pragma unroll
    for (int k = 0; k < G; k++) a[k] = (base16 + (uint32_t) k * 16u + ((uint32_t) (cw >> (4 * k)) & 15u)) * unit;
    constexpr int NG = V == 0 ? 2 : (G <= 4 ? 4 : (G == 5 ? 3 : 2));
    int q = 0;
    for (; q + NG <= n_groups; q += NG) {
        prefilter_trip<G, NG, FB, ASM, PROG>(A, W, lds4, a, first_group + q);
#pragma unroll
        for (int k = 0; k < G; k++) a[k] += NG * G * 16 * unit;
    }
    const int rem = n_groups - q;                       // wave-uniform
    if (NG > 3 && rem == 3) prefilter_trip<G, 3, FB, ASM, PROG>(A, W, lds4, a, first_group + q);
    else if (NG > 2 && rem == 2) prefilter_trip<G, 2, FB, ASM, PROG>(A, W, lds4, a, first_group + q);
    else if (rem == 1) prefilter_trip<G, 1, FB, ASM, PROG>(A, W, lds4, a, first_group + q);
}

#define MS_PF_CASE(GG)                                                                                      \
    case GG:                                                                                                \
        if (fb == 10) prefilter_class<GG, V, 10>(A, lds4, base16, nq, first_group, cw, W);                  \
        else prefilter_class<GG, V, 16>(A, lds4, base16, nq, first_group, cw, W);                           \
        break;

// grid = (blocks per tile, tiles).  One block per CU (the tile's tables fill the LDS), NT/64 waves,
// each wave takes 64 consecutive window starts per iteration.  Dynamic LDS = tables of the
// largest tile, then one queue of kWqCap candidates per wave.
template <int NT, int V, int MW>
__global__ void __launch_bounds__(NT, MW) prefilter_kernel(const PfArgs A) {
    extern __shared__ uint4 lds4[];
    const TileDesc *__restrict__ T = A.tiles + blockIdx.y;
    const uint32_t len16 = T->table_len16;
    const uint4 *__restrict__ src = A.tables + T->table_off16;
    for (uint32_t i = threadIdx.x; i < len16; i += NT) lds4[i] = src[i];
    __syncthreads();
    const int n_classes = T->n_classes;
    PfWave W;
    W.wbuf = reinterpret_cast<uint64_t *>(lds4 + A.wq_off16) + (threadIdx.x >> 6) * kWqCap;
    W.n = 0;
    const int64_t n_chunks = (A.n_bases + NT - 1) / NT;
    // optional clock stamps (measurement runs only; written to a buffer nothing else reads):
    // shader cycles (s_memtime) against the 100 MHz constant clock (s_memrealtime)
    unsigned long long t0 = 0, r0 = 0;
    if (A.clk) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

    // the sequence words of the NEXT chunk are fetched while the current one is scanned
    uint64_t cw_next = 0;
    {
        const int64_t g0 = (int64_t) blockIdx.x * NT + threadIdx.x;
        if (blockIdx.x < n_chunks) cw_next = code_window(A.codes, g0 < A.n_bases ? g0 : 0);
    }
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        W.g = chunk * NT + threadIdx.x;
        W.live = W.g < A.n_bases;
        const uint64_t cw = cw_next;
        {
            const int64_t gn = W.g + (int64_t) gridDim.x * NT;
            if (chunk + gridDim.x < n_chunks) cw_next = code_window(A.codes, gn < A.n_bases ? gn : 0);
        }

        // (walking the classes in a per-wave rotated order, so that the waves do not reach the class
        // boundaries together, measured 5 % SLOWER: the waves then execute 11 different loops at once)
        int c = 0;
        ClassDesc cur = T->cls[c];
        for (int i = 0; i < n_classes; i++) {
            c = c + 1 < n_classes ? c + 1 : 0;
            const ClassDesc nxt = T->cls[c];                      // scalar loads land while this class runs
            const int G = cur.G;
            const int nq = cur.n_groups;
            const int fb = cur.fb;
            const uint32_t base16 = cur.base16;
            const int32_t first_group = cur.first_group;
            switch (G) {
                MS_PF_CASE(1) MS_PF_CASE(2) MS_PF_CASE(3) MS_PF_CASE(4)
                MS_PF_CASE(5) MS_PF_CASE(6) MS_PF_CASE(7) MS_PF_CASE(8)
                MS_PF_CASE(9) MS_PF_CASE(10) MS_PF_CASE(11) MS_PF_CASE(12)
                MS_PF_CASE(13) MS_PF_CASE(14) MS_PF_CASE(15) MS_PF_CASE(16)
                default: break;
            }
            cur = nxt;
        }
    }
    if (W.n > 0) wq_flush(W.wbuf, W.n, A.cand, A.n_cand, A.cand_cap);
    if (A.clk && threadIdx.x == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        const size_t b = (size_t) blockIdx.y * gridDim.x + blockIdx.x;
        A.clk[2 * b] = t1 - t0;
        A.clk[2 * b + 1] = r1 - r0;
    }
}

// ------------------------------------------------------- pre-filter on the matrix cores --
//
// Engine 1 (ms_internal.h, "engine 1"): the same rigorous upper-bound test as above, evaluated as an
// int8 matrix product.  Per wave and iteration: 64 consecutive window starts = two 32-column
// B operands per k-block (the one-hot image of the lane's bases, built once in registers and reused
// by every row tile), and per row tile of 16 motifs x {fwd, rev} one ds_read_b128 per k-block for the
// A operand.  acc >= 0 (sign bit clear) in any of the 16 result registers of a lane marks a candidate.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// 4 bases (8 bits of 2-bit codes) -> 4 words, byte `code` of each word = 1
__device__ __forceinline__ i32x4 onehot4(uint32_t codes8) {
    i32x4 r;
    r.x = (int) (1u << ((codes8 & 3u) << 3));
    r.y = (int) (1u << (((codes8 >> 2) & 3u) << 3));
    r.z = (int) (1u << (((codes8 >> 4) & 3u) << 3));
    r.w = (int) (1u << (((codes8 >> 6) & 3u) << 3));
    return r;
}

// engine 2: 5 bases (10 bits of 2-bit codes) -> bytes 3j, 3j+1, 3j+2 = s1, s2, s1*s2 of base j (s1 = -1 if bit 0 of the
// code is set, s2 = -1 if bit 1 is); byte 15 (the spare k-slot) is filled in by the reader
__device__ __forceinline__ uint4 walsh5(uint32_t codes10) {
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const uint32_t code = (codes10 >> (2 * j)) & 3u;
        const int s1 = (code & 1u) ? -1 : 1, s2 = (code & 2u) ? -1 : 1;
        const int v[3] = {s1, s2, s1 * s2};
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int byte = 3 * j + t;
            w[byte >> 2] |= ((uint32_t) v[t] & 0xFFu) << (8 * (byte & 3));
        }
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ int max16(const i32x16 &c) {
    int a = max(max(c[0], c[1]), c[2]);             // v_max3_i32
    int b = max(max(c[3], c[4]), c[5]);
    int d = max(max(c[6], c[7]), c[8]);
    int e = max(max(c[9], c[10]), c[11]);
    int f = max(max(c[12], c[13]), c[14]);
    a = max(max(a, b), d);
    e = max(max(e, f), c[15]);
    return max(a, e);
}

// bit n of the result = result register 15 - n is non-negative (field n of the lane's table group)
__device__ __forceinline__ uint32_t nonneg_flags(const i32x16 &c) {
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) m = __builtin_amdgcn_alignbit(m, (uint32_t) c[j], 31);   // (m << 1) | sign
    return ~m & 0xFFFFu;
}

struct MfWave {
    uint64_t *wbuf;
    uint32_t n;
};

__device__ __forceinline__ void emit_rec(const PfArgs &A, MfWave &W, bool live, int64_t g, uint32_t flags, int32_t group) {
    const bool flagged = live && flags != 0;
    const unsigned long long mask = __ballot(flagged);
    if (mask == 0) return;
    const uint32_t n_new = (uint32_t) __popcll(mask);
    if (W.n + n_new > (uint32_t) kWqCap) {
        wq_flush(W.wbuf, W.n, A.cand, A.n_cand, A.cand_cap);
        W.n = 0;
    }
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
    if (flagged) W.wbuf[W.n + rank] = cand_pack((uint64_t) g, (uint32_t) group, flags);
    W.n += n_new;
}

// the record without flags (flags = 0): expand_kernel recomputes the group's 16 row sums for the flagged lanes
__device__ __forceinline__ void emit_rec_noflags(const PfArgs &A, MfWave &W, bool flagged, uint64_t gkey, int32_t group) {
    const unsigned long long mask = __ballot(flagged);
    if (mask == 0) return;
    const uint32_t n_new = (uint32_t) __popcll(mask);
    if (W.n + n_new > (uint32_t) kWqCap) {
        wq_flush(W.wbuf, W.n, A.cand, A.n_cand, A.cand_cap);
        W.n = 0;
    }
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
    if (flagged) W.wbuf[W.n + rank] = gkey | ((uint64_t) (uint32_t) group << 16);
    W.n += n_new;
}

// any of the 32 result registers of the two 32-window operands non-negative?  (16 x v_max3_i32)
__device__ __forceinline__ int max32(const i32x16 &c, const i32x16 &d) {
    int m[11];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        m[i] = max(max(c[3 * i], c[3 * i + 1]), c[3 * i + 2]);
        m[5 + i] = max(max(d[3 * i], d[3 * i + 1]), d[3 * i + 2]);
    }
    m[10] = max(max(c[15], d[15]), m[0]);
    const int x = max(max(m[1], m[2]), m[3]), y = max(max(m[4], m[5]), m[6]), z = max(max(m[7], m[8]), m[9]);
    return max(max(x, y), max(z, m[10]));
}

// Rare path (about one tile in five has a candidate in some lane): which of the two 32-window operands, which
// fields; queue the records.  (Inlined: a real call would pass the 32 result registers and the argument block
// through scratch memory.)
__device__ __forceinline__ void mfma_emit(const PfArgs &A, MfWave &W, const i32x16 &c0, const i32x16 &c1, int32_t group,
                                          int64_t g0, bool live0, bool live1) {
    if (__any(max16(c0) >= 0)) emit_rec(A, W, live0, g0, nonneg_flags(c0), group);
    if (__any(max16(c1) >= 0)) emit_rec(A, W, live1, g0 + 32, nonneg_flags(c1), group);
}

// All row tiles of one class (NK k-blocks each): per tile NK ds_read_b128 (A operand), 2 * NK matrix
// instructions, 16 v_max3 and one compare.  The B operands (one-hot image of the lane's bases) come from a
// 256-entry table in LDS: 4 bases (one byte of 2-bit codes) -> 16 operand bytes.
template <int NK, int V, int ENG, bool MEAS>
__device__ __forceinline__ void mfma_class(const PfArgs &A, MfWave &W, const char *__restrict__ lds, const char *__restrict__ lut,
                                           uint32_t byte_off, int n_row_tiles, int32_t first_group, uint64_t cw0, uint64_t cw1,
                                           int64_t g0, bool live0, bool live1) {
    const uint32_t lane = threadIdx.x & 63u, h = lane >> 5;
    const char *p = lds + byte_off + lane * 16u;
    constexpr int kStep = NK * kMfmaRowTileBytesPerKb;
    i32x4 b0[NK], b1[NK];
#pragma unroll
    for (int kb = 0; kb < NK; kb++) {
        if constexpr (ENG == 2) {
            // 5 bases per lane half; the spare k-slot (byte 15) carries the bias scale: 64 in half 0, 1 in half 1
            const int spare = h ? (1 << 24) : (64 << 24);
            b0[kb] = *reinterpret_cast<const i32x4 *>(lut + (((uint32_t) (cw0 >> (20 * kb + 10 * h)) & 0x3FFu) << 4));
            b1[kb] = *reinterpret_cast<const i32x4 *>(lut + (((uint32_t) (cw1 >> (20 * kb + 10 * h)) & 0x3FFu) << 4));
            b0[kb].w |= spare;
            b1[kb].w |= spare;
        } else {
            b0[kb] = *reinterpret_cast<const i32x4 *>(lut + (((uint32_t) (cw0 >> (16 * kb + 8 * h)) & 0xFFu) << 4));
            b1[kb] = *reinterpret_cast<const i32x4 *>(lut + (((uint32_t) (cw1 >> (16 * kb + 8 * h)) & 0xFFu) << 4));
        }
    }
    const i32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto product = [&](const char *q, i32x16 &c0, i32x16 &c1) {
        i32x4 a[NK];
#pragma unroll
        for (int kb = 0; kb < NK; kb++) a[kb] = *reinterpret_cast<const i32x4 *>(q + kb * kMfmaRowTileBytesPerKb);
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], b0[0], z, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], b1[0], z, 0, 0, 0);
#pragma unroll
        for (int kb = 1; kb < NK; kb++) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[kb], b0[kb], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[kb], b1[kb], c1, 0, 0, 0);
        }
    };
    auto test = [&](const i32x16 &c0, const i32x16 &c1, int t) {
        if constexpr (V & 4) {
            // default: the rare path only queues (position, table group) of the lanes that flagged anything; which of the group's
            // 16 fields it was is recomputed by expand_kernel from the same int8 tables (ms_tail.hip)
            const int m0 = max16(c0), m1 = max16(c1);
            if (__builtin_expect(__any(max(m0, m1) >= 0) && !(MEAS && A.no_emit), 0)) {
                const int32_t group = first_group + 2 * t + (int32_t) h;
                emit_rec_noflags(A, W, live0 && m0 >= 0, (uint64_t) g0 << 30, group);
                emit_rec_noflags(A, W, live1 && m1 >= 0, (uint64_t) (g0 + 32) << 30, group);
            }
        } else if constexpr (V & 2) {
            // A/B: the two halves' maxima are kept (17 instead of 16 max instructions per tile), so the rare path does not
            // recompute them (2 x 8 fewer there)
            const int m0 = max16(c0), m1 = max16(c1);
            if (__builtin_expect(__any(max(m0, m1) >= 0) && !(MEAS && A.no_emit), 0)) {
                const int32_t group = first_group + 2 * t + (int32_t) h;
                if (__any(m0 >= 0)) emit_rec(A, W, live0, g0, nonneg_flags(c0), group);
                if (__any(m1 >= 0)) emit_rec(A, W, live1, g0 + 32, nonneg_flags(c1), group);
            }
        } else {
            if (__builtin_expect(__any(max32(c0, c1) >= 0) && !(MEAS && A.no_emit), 0))
                mfma_emit(A, W, c0, c1, first_group + 2 * t + (int32_t) h, g0, live0, live1);
        }
    };
    // ILP tiles' products are issued back to back (independent accumulators), then reduced: a wave that
    // spends more of its time issuing matrix instructions leaves the pipe idle less often (4 waves per SIMD)
    constexpr int ILP = (V & 1) && NK <= 2 ? 2 : 1;      // V: bit 0 two tiles in flight, bit 1 per-half maxima (A/B), bit 2 flag-free records
    int t = 0;
    for (; t + ILP <= n_row_tiles; t += ILP, p += ILP * kStep) {
        i32x16 c0[ILP], c1[ILP];
#pragma unroll
        for (int u = 0; u < ILP; u++) product(p + u * kStep, c0[u], c1[u]);
        if constexpr ((V & 8) && ILP == 2) {
            // A/B: ONE branch for the pair of tiles (the vector -> scalar -> branch chain is paid once per two tiles)
            const int ma = max32(c0[0], c1[0]), mb = max32(c0[1], c1[1]);
            if (__builtin_expect(__any(max(ma, mb) >= 0) && !(MEAS && A.no_emit), 0)) {
                if (__any(ma >= 0)) mfma_emit(A, W, c0[0], c1[0], first_group + 2 * t + (int32_t) h, g0, live0, live1);
                if (__any(mb >= 0)) mfma_emit(A, W, c0[1], c1[1], first_group + 2 * (t + 1) + (int32_t) h, g0, live0, live1);
            }
        } else {
#pragma unroll
            for (int u = 0; u < ILP; u++) test(c0[u], c1[u], t + u);
        }
    }
    for (; t < n_row_tiles; t++, p += kStep) {
        i32x16 c0, c1;
        product(p, c0, c1);
        test(c0, c1, t);
    }
}

// ---- engine 3: the same tiles on v_mfma_scale_f32_32x32x64_f8f6f4 (A fp6 e2m3, B fp4 one-hot), 16 motif columns per k-block
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ i32x16 as_bits(const f32x16 &c) {           // the f32 results as bit patterns: sign bit clear <=> value >= +0
    i32x16 r;
#pragma unroll
    for (int j = 0; j < 16; j++) r[j] = __float_as_int(c[j]);
    return r;
}

template <int NK, int V, bool MEAS>
__device__ __forceinline__ void mfma_class_f6(const PfArgs &A, MfWave &W, const char *__restrict__ lds, const char *__restrict__ lut,
                                              uint32_t byte_off, int n_row_tiles, int32_t first_group, uint64_t cw0, uint64_t cw1,
                                              int64_t g0, bool live0, bool live1) {
    const uint32_t lane = threadIdx.x & 63u, h = lane >> 5;
    const char *p = lds + byte_off + lane * 8u;
    constexpr int kStep = NK * kF6BytesPerKb;
    // B operands: the lane's 8 bases of k-block kb (columns 16 kb + 8 h ...) as 32 fp4 one-hot k-slots = two table reads of 8 bytes
    i32x8 b0[NK], b1[NK];
#pragma unroll
    for (int kb = 0; kb < NK; kb++) {
        const uint32_t c0 = (uint32_t) (cw0 >> (32 * kb + 16 * h)) & 0xFFFFu, c1 = (uint32_t) (cw1 >> (32 * kb + 16 * h)) & 0xFFFFu;
        const int2 l0 = *reinterpret_cast<const int2 *>(lut + ((c0 & 0xFFu) << 3)), h0 = *reinterpret_cast<const int2 *>(lut + ((c0 >> 8) << 3));
        const int2 l1 = *reinterpret_cast<const int2 *>(lut + ((c1 & 0xFFu) << 3)), h1 = *reinterpret_cast<const int2 *>(lut + ((c1 >> 8) << 3));
        b0[kb] = i32x8{l0.x, l0.y, h0.x, h0.y, 0, 0, 0, 0};
        b1[kb] = i32x8{l1.x, l1.y, h1.x, h1.y, 0, 0, 0, 0};
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
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], b0[0], z, 2, 4, 0, 127, 0, 127);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], b1[0], z, 2, 4, 0, 127, 0, 127);
#pragma unroll
        for (int kb = 1; kb < NK; kb++) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[kb], b0[kb], c0, 2, 4, 0, 127, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[kb], b1[kb], c1, 2, 4, 0, 127, 0, 127);
        }
    };
    auto test = [&](const f32x16 &f0, const f32x16 &f1, int t) {
        const i32x16 c0 = as_bits(f0), c1 = as_bits(f1);
        if constexpr (V & 2) {                                // the two halves' maxima are kept for the rare path (this engine is VALU-issue bound)
            const int m0 = max16(c0), m1 = max16(c1);
            if (__builtin_expect(__any(max(m0, m1) >= 0) && !(MEAS && A.no_emit), 0)) {
                const int32_t group = first_group + 2 * t + (int32_t) h;
                if (__any(m0 >= 0)) emit_rec(A, W, live0, g0, nonneg_flags(c0), group);
                if (__any(m1 >= 0)) emit_rec(A, W, live1, g0 + 32, nonneg_flags(c1), group);
            }
        } else {
            if (__builtin_expect(__any(max32(c0, c1) >= 0) && !(MEAS && A.no_emit), 0))
                mfma_emit(A, W, c0, c1, first_group + 2 * t + (int32_t) h, g0, live0, live1);
        }
    };
    constexpr int ILP = (V & 1) ? 2 : 1;
    int t = 0;
    for (; t + ILP <= n_row_tiles; t += ILP, p += ILP * kStep) {
        f32x16 c0[ILP], c1[ILP];
#pragma unroll
        for (int u = 0; u < ILP; u++) product(p + u * kStep, c0[u], c1[u]);
        // keep BOTH tiles' matrix instructions ahead of the first reduction (left alone, hipcc sinks the second tile's below
        // the first tile's test and the wave sits out its own result latency once per tile)
        if constexpr (ILP > 1) {
#pragma unroll
            for (int u = 0; u < ILP; u++) asm volatile("" : "+v"(c0[u]), "+v"(c1[u]));      // (an empty asm "uses" the results here: the IR-level sinking cannot pass it)
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < ILP; u++) test(c0[u], c1[u], t + u);
    }
    for (; t < n_row_tiles; t++, p += kStep) {
        f32x16 c0, c1;
        product(p, c0, c1);
        test(c0, c1, t);
    }
}

// grid = (blocks per tile, tiles); NT / 64 waves per block, each takes 64 consecutive window starts
// per iteration (lanes l and l + 32 share window l & 31 and hold the two halves of every k-block).
// Dynamic LDS: operand tables of the tile | wave queues | B-operand table (kMfmaLutBytes / kMfma2LutBytes).
// V (A/B measurement): 1 = two row tiles' products in flight per wave in the narrow classes.
// (Measured and dropped, tools/pf_variants.py: fetching the next chunk's sequence words early, class descriptors
// in registers, waves walking the classes in rotated order, tiles software-pipelined in pairs -- each within noise;
// 5 waves per SIMD (two 640-thread blocks per CU at <= 96 VGPRs) spills and is 35 % slower; software-pipelining the
// one-k-block class alone: 66 -> 63 cycles per matrix instruction on an all-W=8 set, < 1 % on the benchmark set;
// s_setprio raised around the matrix instructions: within noise; 12 waves per CU: +7 % time at +3 % clock;
// A operands fetched one row tile ahead (first fetch before the class's B operands are waited for): +4 % time;
// 128 windows per wave in the narrow classes (each A operand serves four B operands): 64 + 32 + 8 registers of tiles
// do not fit 128 VGPRs, 80 spills, +70 % time.)
// MEAS: the measurement-only instantiation (drop candidates, clock stamps); the product kernel carries neither.
// MAXNK / WPS (A/B): a kernel that only knows row tiles of <= MAXNK k-blocks needs fewer registers (B operands: 8 per k-block),
// WPS = waves per SIMD the register allocation must leave room for (two 768-thread blocks per CU = 6).
// HANDOUT: 0 = a BLOCK takes 4 chunks of NT positions per atomic behind two __syncthreads (every wave then waits for the block's
// slowest: -10 % with 16 waves per block, -2 % with 8); >= 1 = every WAVE takes its own units, no barrier in the loop -- the shipped
// engine-3 form (two 512-thread blocks per CU, variant 46; profiles/r02_wave_occupancy_ab.log).
template <int NT, int V, int ENG, bool MEAS, int MAXNK = 4, int WPS = NT / 256, int HANDOUT = 0>
__global__ void __launch_bounds__(NT, WPS) prefilter_mfma_kernel(const PfArgs A) {
    extern __shared__ uint4 lds4[];
    const TileDesc *__restrict__ T = A.tiles + blockIdx.y;
    const uint32_t len16 = T->table_len16;
    const uint4 *__restrict__ src = A.tables + T->table_off16;
    for (uint32_t i = threadIdx.x; i < len16; i += NT) lds4[i] = src[i];
    uint4 *lut4 = lds4 + A.wq_off16 + kWqBytes / 16;
    if constexpr (ENG == 3) {
        // byte of four 2-bit codes -> 16 fp4 k-slots (8 bytes): slot 4 c + code_c = 1.0 (e2m1 code 0x2)
        uint2 *lut2 = reinterpret_cast<uint2 *>(lut4);
        for (uint32_t i = threadIdx.x; i < 256u; i += NT) {
            unsigned long long w = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) w |= 2ULL << (4 * (4 * c + (int) ((i >> (2 * c)) & 3u)));
            lut2[i] = make_uint2((uint32_t) w, (uint32_t) (w >> 32));
        }
    } else if constexpr (ENG == 2) {
        for (uint32_t i = threadIdx.x; i < 1024u; i += NT) lut4[i] = walsh5(i);
    } else {
        for (uint32_t i = threadIdx.x; i < 256u; i += NT) {
            const i32x4 v = onehot4(i);
            lut4[i] = make_uint4((uint32_t) v.x, (uint32_t) v.y, (uint32_t) v.z, (uint32_t) v.w);
        }
    }
    __syncthreads();
    const char *lds = reinterpret_cast<const char *>(lds4);
    const char *lut = reinterpret_cast<const char *>(lut4);
    const int n_classes = T->n_classes;
    MfWave W;
    W.wbuf = reinterpret_cast<uint64_t *>(lds4 + A.wq_off16) + (threadIdx.x >> 6) * kWqCap;
    W.n = 0;
    const uint32_t lane = threadIdx.x & 63u, r = lane & 31u;
    const int64_t n_chunks = (A.n_bases + NT - 1) / NT;
    unsigned long long t0 = 0, r0 = 0;
    if constexpr (MEAS) { if (A.clk) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); } }

    // Work is handed out DYNAMICALLY (per wave or per block, HANDOUT): the kernel wants every CU whole (all its LDS and registers),
    // so a block whose CU is still busy with another stream's kernel (an upload's pack, a copy-out's blit: the batch stream runs
    // them beside the scan) starts late -- with a static partition the whole launch then waits for that block's full share
    // (measured: 3.7x on the streamed sweep with copy-out, profiles/r02_stream_coexistence.log); now it simply takes less.
    auto scan_pass = [&](int64_t g0) {                                       // 64 window starts of this wave against every class
        const bool live0 = g0 < A.n_bases, live1 = g0 + 32 < A.n_bases;
        const uint64_t cw0 = code_window(A.codes, live0 ? g0 : 0);
        const uint64_t cw1 = code_window(A.codes, live1 ? g0 + 32 : 0);
        for (int i = 0; i < n_classes; i++) {
            const ClassDesc cd = T->cls[i];
            const uint32_t off = cd.base16 * 16u;
            if constexpr (ENG == 3) {
                if (cd.G == 1) mfma_class_f6<1, V, MEAS>(A, W, lds, lut, off, cd.n_groups, cd.first_group, cw0, cw1, g0, live0, live1);
                else if (cd.G == 2) mfma_class_f6<2, V, MEAS>(A, W, lds, lut, off, cd.n_groups, cd.first_group, cw0, cw1, g0, live0, live1);
                continue;
            }
            switch (cd.G) {
                case 1: mfma_class<1, V, ENG, MEAS>(A, W, lds, lut, off, cd.n_groups, cd.first_group, cw0, cw1, g0, live0, live1); break;
                case 2: mfma_class<2, V, ENG, MEAS>(A, W, lds, lut, off, cd.n_groups, cd.first_group, cw0, cw1, g0, live0, live1); break;
                case 3: if constexpr (MAXNK >= 3) mfma_class<3, V, ENG, MEAS>(A, W, lds, lut, off, cd.n_groups, cd.first_group, cw0, cw1, g0, live0, live1); break;
                case 4: if constexpr (ENG == 1 && MAXNK >= 4) mfma_class<4, V, ENG, MEAS>(A, W, lds, lut, off, cd.n_groups, cd.first_group, cw0, cw1, g0, live0, live1); break;
                default: break;
            }
        }
    };
    if constexpr (HANDOUT >= 1) {
        // Per-WAVE hand-out: a wave takes A.wave_passes x 64 consecutive window starts per atomic (the next unit is requested
        // before the current one is scanned, so the atomic's latency is hidden) and never meets the block's other waves again: no
        // barrier in the loop, a wave that ran into the rare path more often than its neighbours delays nobody.  The host sizes
        // the unit so that the waves of a counter word stay below ~50 atomics per microsecond on it (a word saturates near 90: with ONE
        // word per tile, 4 passes per atomic on the 579-motif set cost +13 %); an input worth few units per wave is split evenly.
        const uint32_t wave_passes = A.wave_passes < 1 ? (uint32_t) HANDOUT : (uint32_t) A.wave_passes;
        const uint32_t n_passes_total = (uint32_t) ((A.n_bases + 63) / 64);           // <= 2^28: a set holds <= 2^34 bases
        const uint32_t n_units = (n_passes_total + wave_passes - 1) / wave_passes;
        constexpr uint32_t wpb = NT / 64;
        if (n_units <= gridDim.x * wpb) {                                             // a small input: one unit per wave, no atomic at all
            const uint32_t unit = blockIdx.x * wpb + (threadIdx.x >> 6);
            if (unit < n_units)
                for (uint32_t j = 0; j < wave_passes; j++) scan_pass((int64_t) (unit * wave_passes + j) * 64 + r);
        } else {
            // kPfCounters counter words per tile, 64 bytes apart: the blocks are dealt round-robin onto them and a word hands out every
            // kPfCounters-th unit, so that the units can be small (a short tail: the launch ends one unit after its last wave starts
            // one) without the words saturating (~90 atomics per microsecond each).  A wave's first unit in its group is its own
            // number there; the words start at 0 and the waves add their group's size themselves.
            const uint32_t K = gridDim.x < (uint32_t) kPfCounters ? gridDim.x : (uint32_t) kPfCounters;     // every word needs a block
            const uint32_t g = blockIdx.x % K;
            const uint32_t waves_g = ((gridDim.x - g + K - 1) / K) * wpb;
            const uint32_t units_g = n_units > g ? (n_units - g + K - 1) / K : 0u;
            unsigned int *word = A.chunk_counter + ((size_t) blockIdx.y * kPfCounters + g) * 16;
            auto take = [&]() {
                unsigned int u = 0;
                if (lane == 0) u = atomicAdd(word, 1u);
                return waves_g + (uint32_t) __builtin_amdgcn_readfirstlane((int) u);
            };
            uint32_t v = (blockIdx.x / K) * wpb + (threadIdx.x >> 6);
            while (v < units_g) {
                const uint32_t next = take();                                         // asked for before this unit is scanned
                const uint32_t p0 = (v * K + g) * wave_passes;
                for (uint32_t j = 0; j < wave_passes; j++)                            // passes past the end scan dead lanes (last unit only)
                    scan_pass((int64_t) (p0 + j) * 64 + r);
                v = next;
            }
        }
    } else {
    constexpr int kSuper = 4;
    __shared__ unsigned int s_super;
    for (;;) {
        __syncthreads();                                                   // every wave is done with the previous hand-out
        if (threadIdx.x == 0) s_super = atomicAdd(A.chunk_counter + blockIdx.y, 1u);
        __syncthreads();
        const int64_t first = (int64_t) s_super * kSuper;
        if (first >= n_chunks) break;
        for (int64_t chunk = first; chunk < first + kSuper && chunk < n_chunks; chunk++)
            scan_pass(chunk * NT + (threadIdx.x & ~63u) + r);              // window start of N-tile 0; N-tile 1: + 32
    }
    }
    if (W.n > 0) wq_flush(W.wbuf, W.n, A.cand, A.n_cand, A.cand_cap);
    if constexpr (MEAS) {
        if (A.clk && threadIdx.x == 0) {
            const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
            const size_t b = (size_t) blockIdx.y * gridDim.x + blockIdx.x;
            A.clk[2 * b] = t1 - t0;
            A.clk[2 * b + 1] = r1 - r0;
        }
    }
}

// -------------------------------------------------------------------- fp64 kernels --

// Windows that overlap a non-ACGT base are scored in fp64 outright: the pre-filter packs such
// bases as 'A', so its answer for these windows means nothing (and rescore_kernel skips them).
// Two steps so that the rare work is spread over the whole chip instead of a few waves:
//   nlist_kernel  one thread per 32 positions: list the positions whose next max_w bases hold an N
//   neval_kernel  one thread per (listed position, chunk of kNwMotifChunk motifs)
__global__ void __launch_bounds__(256) nlist_kernel(const uint32_t *__restrict__ nmask, int64_t n_bases, int max_w,
                                                    NPos *__restrict__ list, unsigned long long *__restrict__ n_list,
                                                    uint64_t cap) {
    const int64_t n_words = (n_bases + 31) / 32;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    const uint32_t wm = low_mask(max_w);
    for (int64_t j = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; j < n_words; j += stride) {
        const uint32_t w0 = nmask[j], w1 = nmask[j + 1];
        if ((w0 | w1) == 0) continue;
        const uint64_t comb = ((uint64_t) w1 << 32) | w0;
        uint32_t qual = 0;                                   // positions of this word whose window holds an N
        for (int b = 0; b < 32; b++)
            if (j * 32 + b < n_bases && ((uint32_t) (comb >> b) & wm) != 0) qual |= 1u << b;
        if (qual == 0) continue;
        unsigned long long i = atomicAdd(n_list, (unsigned long long) __popc(qual));
        while (qual) {
            const int b = __ffs((int) qual) - 1;
            qual &= qual - 1u;
            if (i < cap) list[i].g = j * 32 + b;
            i++;
        }
    }
}

// everything about a listed position that does not depend on the motif, computed once
__global__ void __launch_bounds__(256) nprep_kernel(const DevSeq S, NPos *__restrict__ list, const unsigned long long *__restrict__ n_list,
                                                    uint64_t cap, const HitOut H) {
    unsigned long long n = *n_list;
    if (n > cap) n = cap;
    for (unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long) gridDim.x * blockDim.x) {
        const int64_t g = list[i].g;
        const int64_t r = find_region(S, g);
        const int64_t room = S.offsets[r + 1] - g;
        NPos q;
        q.g = g;
        q.cw = code_window(S.codes, g);
        q.nw = n_window(S.nmask, g);
        q.coord = hit_coord(H, S, r, g);
        q.room = (int32_t) (room < 64 ? room : 64);
        list[i] = q;
    }
}

__global__ void __launch_bounds__(256) neval_kernel(const DevSeq S, const DevPwm Pw, const int32_t *__restrict__ motifs,
                                                    int32_t n_motifs, int strand_mask, const NPos *__restrict__ list,
                                                    const unsigned long long *__restrict__ n_list, uint64_t cap,
                                                    const HitOut H) {
    // the block's motifs never change: their fp64 tables are read from LDS, not through L1/L2
    __shared__ double2 s_tab[kNwMotifChunk * kMaxFastWidth * 4];
    __shared__ int32_t s_motif[kNwMotifChunk], s_width[kNwMotifChunk], s_off[kNwMotifChunk];
    unsigned long long n = *n_list;
    if (n > cap) n = cap;
    const int m0 = blockIdx.y * kNwMotifChunk;
    const int cnt = min(kNwMotifChunk, n_motifs - m0);
    __shared__ int64_t s_src[kNwMotifChunk];
    if ((int) threadIdx.x < cnt) {
        const int32_t p = motifs[m0 + threadIdx.x];
        s_motif[threadIdx.x] = p;
        s_width[threadIdx.x] = Pw.width[p];
        s_src[threadIdx.x] = Pw.tab_off[p];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int off = 0;
        for (int m = 0; m < cnt; m++) { s_off[m] = off; off += s_width[m] * 4; }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < cnt * kMaxFastWidth * 4; j += blockDim.x) {       // all motifs' loads in flight together
        const int m = j / (kMaxFastWidth * 4), i = j % (kMaxFastWidth * 4);
        if (i < s_width[m] * 4) s_tab[s_off[m] + i] = Pw.tab2[s_src[m] + i];
    }
    __syncthreads();
    // Upper bound of a window's raw score when one contiguous run of columns [a, b) is non-ACGT (adds nothing, cscore.c:345-353):
    // the best base of every other column = pre[a] + suf[b].  Most windows that overlap a run of N lose too many columns to
    // reach the cutoff; they are dismissed by two table reads instead of 2 W fp64 adds.
    __shared__ double s_pre[kNwMotifChunk][2][kMaxFastWidth + 1], s_suf[kNwMotifChunk][2][kMaxFastWidth + 1];
    __shared__ double s_floor[kNwMotifChunk];
    if ((int) threadIdx.x < 2 * cnt) {
        const int m = threadIdx.x >> 1, sd = threadIdx.x & 1, W = s_width[m];
        const double2 *t = s_tab + s_off[m];
        double acc = 0.0;
        s_pre[m][sd][0] = 0.0;
        for (int c = 0; c < W; c++) {
            double hi = -INFINITY;
            for (int b = 0; b < 4; b++) hi = fmax(hi, sd ? t[c * 4 + b].y : t[c * 4 + b].x);
            acc += hi;
            s_pre[m][sd][c + 1] = acc;
        }
        acc = 0.0;
        s_suf[m][sd][W] = 0.0;
        for (int c = W - 1; c >= 0; c--) {
            double hi = -INFINITY;
            for (int b = 0; b < 4; b++) hi = fmax(hi, sd ? t[c * 4 + b].y : t[c * 4 + b].x);
            acc += hi;
            s_suf[m][sd][c] = acc;
        }
        if (sd == 0) {                                          // the raw-sum floor of the hit test (ms_api.hip), minus room for this bound's own rounding
            const double fl = Pw.raw_floor[s_motif[m]];
            s_floor[m] = fl - 1e-9 * (1.0 + fabs(fl));
        }
    }
    __syncthreads();
    // Two phases per round of 256 positions, so that the fp64 scoring runs DENSE: (1) every lane checks its position against the
    // block's 8 motifs (does the window reach an N, does it fit its region, can its non-N columns reach the cutoff) and queues
    // the few (position, motif) pairs that survive; (2) the lanes take one queued pair each.  Scoring inline would make every wave
    // execute the scoring path for every motif as soon as ONE of its 64 positions needs it (measured: 0.35 -> see DESIGN.md).
    constexpr int U = 1;                                  // positions per lane and round (4 with a quarter of the blocks measured 20-70 % slower: the kernel wants many small blocks)
    constexpr unsigned int kWorkCap = 4096;
    __shared__ uint16_t s_work[kWorkCap];                 // (position slot in the round: 10 bits) << 3 | motif of the block
    __shared__ unsigned int s_nwork;
    const unsigned long long per_sub = (unsigned long long) gridDim.x * blockDim.x;
    const unsigned long long per_round = per_sub * U;
    const unsigned long long rounds = (n + per_round - 1) / per_round;
    for (unsigned long long rd = 0; rd < rounds; rd++) {
        if (threadIdx.x == 0) s_nwork = 0;
        __syncthreads();
        const unsigned long long i0 = rd * per_round + (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
        NPos q[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            q[u].nw = 0;                                    // no N: no motif's window reaches one
            if (i0 + u * per_sub < n) q[u] = list[i0 + u * per_sub];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (q[u].nw == 0) continue;
            for (int m = 0; m < cnt; m++) {
                const int W = s_width[m];                           // <= 32: only pre-filter motifs come here
                const uint32_t nm = q[u].nw & low_mask(W);
                if (nm == 0) continue;
                if (W > q[u].room) continue;                        // window runs past its region (cscore.c:340)
                const int a = __ffs((int) nm) - 1, b = 32 - __clz((int) nm);
                if (__popc(nm) == b - a) {                           // one contiguous run (the usual case): bound by the columns outside it
                    const double fl = s_floor[m];
                    const bool dead_f = !(strand_mask & 1) || s_pre[m][0][a] + s_suf[m][0][b] < fl;
                    const bool dead_r = !(strand_mask & 2) || s_pre[m][1][a] + s_suf[m][1][b] < fl;
                    if (dead_f && dead_r) continue;
                }
                const unsigned int slot = atomicAdd(&s_nwork, 1u);
                if (slot < kWorkCap) {
                    s_work[slot] = (uint16_t) (((uint32_t) (u * 256 + (int) threadIdx.x) << 3) | (uint32_t) m);
                } else {                                             // queue full (dense N): score here
                    double fwd, rev;
                    score_window32(s_tab + s_off[m], W, q[u].cw, q[u].nw, fwd, rev);
                    test_and_emit(H, Pw, (uint32_t) s_motif[m], q[u].coord, fwd, rev, strand_mask);
                }
            }
        }
        __syncthreads();
        const unsigned int nw_items = s_nwork < kWorkCap ? s_nwork : kWorkCap;
        for (unsigned int k = threadIdx.x; k < nw_items; k += blockDim.x) {
            const uint32_t item = s_work[k];
            const int m = (int) (item & 7u);
            const uint32_t slot = item >> 3;                         // u * 256 + thread
            const NPos p = list[rd * per_round + (unsigned long long) (slot >> 8) * per_sub + (unsigned long long) blockIdx.x * blockDim.x + (slot & 255u)];
            double fwd, rev;
            score_window32(s_tab + s_off[m], s_width[m], p.cw, p.nw, fwd, rev);
            test_and_emit(H, Pw, (uint32_t) s_motif[m], p.coord, fwd, rev, strand_mask);
        }
        __syncthreads();
    }
}

// grid = (ceil(n_bases/256), n_exact motifs).  Fallback for motifs the pre-filter cannot take
// (W > 32, max_raw <= 0, non-finite values, cutoff below the quantiser's floor).
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

// U candidate records per thread and round, their loads issued side by side: the kernel is a chain of dependent gathers
// (record -> region hint / sequence words / motif id -> offsets / width / table offset -> table entries), so the records in
// flight per thread -- not the arithmetic -- set its speed; one barrier pair per round of U records instead of per record.
constexpr int kRescoreU = 4;

__global__ void __launch_bounds__(256) rescore_kernel(const DevSeq S, const DevPwm Pw, const uint64_t *__restrict__ cand,
                                                      const unsigned long long *__restrict__ n_cand, uint64_t cand_cap,
                                                      const int32_t *__restrict__ group_motifs, int strand_mask,
                                                      const HitOut H) {
    __shared__ HitStage st;
    if (threadIdx.x == 0) st.n = 0;
    __syncthreads();
    unsigned long long n = *n_cand;
    if (n > cand_cap) n = cand_cap;
    constexpr int U = kRescoreU;
    const unsigned long long per_sub = (unsigned long long) gridDim.x * blockDim.x;
    const unsigned long long per_round = per_sub * U;
    const unsigned long long rounds = (n + per_round - 1) / per_round;
    for (unsigned long long rd = 0; rd < rounds; rd++) {
        const unsigned long long i0 = rd * per_round + (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
        uint64_t c[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            live[u] = i0 + u * per_sub < n;
            c[u] = live[u] ? cand[i0 + u * per_sub] : 0;
        }
        // independent of each other: region hint, sequence words, N words, first flagged motif
        int64_t g[U], lo[U];
        uint64_t cw[U];
        uint32_t nw[U], flags[U];
        int32_t group[U], pm[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            g[u] = (int64_t) (c[u] >> 30);
            group[u] = (int32_t) ((c[u] >> 16) & 0x3FFFu);
            uint32_t f = (uint32_t) c[u] & 0xFFFFu;                    // bit n = field n; motif slot n >> 1
            flags[u] = (f | (f >> 1)) & 0x5555u;                        // both strands are re-scored anyway
            lo[u] = S.blk2reg[g[u] >> 6];
            cw[u] = code_window(S.codes, g[u]);
            nw[u] = n_window(S.nmask, g[u]);
            pm[u] = flags[u] ? group_motifs[group[u] * kGroupSlots + ((__ffs((int) flags[u]) - 1) >> 1)] : -1;
        }
        // second hop: the region's bounds, the first motif's width / table offset
        int64_t r[U], beg[U], end[U];
        int W[U];
        int64_t toff[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t o0 = S.offsets[lo[u]], o1 = S.offsets[lo[u] + 1];
            const int64_t o2 = lo[u] + 2 <= S.R ? S.offsets[lo[u] + 2] : o1;
            if (g[u] < o1) { r[u] = lo[u]; beg[u] = o0; end[u] = o1; }
            else if (g[u] < o2) { r[u] = lo[u] + 1; beg[u] = o1; end[u] = o2; }
            else { r[u] = find_region(S, g[u]); beg[u] = S.offsets[r[u]]; end[u] = S.offsets[r[u] + 1]; }      // tiny regions
            W[u] = pm[u] >= 0 ? Pw.width[pm[u]] : 0;
            toff[u] = pm[u] >= 0 ? Pw.tab_off[pm[u]] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (!live[u]) continue;
            const int64_t gk = H.pbits ? (int64_t) (((uint64_t) r[u] << H.pbits) | (uint64_t) (g[u] - beg[u])) : g[u];
            bool first = true;
            while (flags[u]) {
                const int slot = (__ffs((int) flags[u]) - 1) >> 1;
                flags[u] &= flags[u] - 1u;
                int32_t m = pm[u];
                int w = W[u];
                int64_t to = toff[u];
                if (!first) {                                            // further motifs of the group: rare
                    m = group_motifs[group[u] * kGroupSlots + slot];
                    if (m >= 0) { w = Pw.width[m]; to = Pw.tab_off[m]; }
                }
                first = false;
                if (m < 0) continue;
                if (g[u] + w > end[u]) continue;                         // window runs past its region (cscore.c:340)
                if (nw[u] & low_mask(w)) continue;                       // scored by neval_kernel
                double fwd, rev;
                score_window32(Pw.tab2 + to, w, cw[u], 0u, fwd, rev);    // no N in the window (checked above)
                test_and_stage(st, H, Pw, (uint32_t) m, gk, fwd, rev, strand_mask);
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

__global__ void __launch_bounds__(256) finalize_kernel(const uint64_t *__restrict__ keys, int64_t n, int gbits, int32_t P,
                                                       const DevSeq S,
                                                       int64_t *__restrict__ seq_idx, int64_t *__restrict__ pos,
                                                       int8_t *__restrict__ strand, int64_t *__restrict__ motif_first,
                                                       unsigned long long *__restrict__ region_counts) {
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
__global__ void __launch_bounds__(256) finalize_rp_kernel(const uint64_t *__restrict__ keys, int64_t n, int rbits, int pbits, int32_t P,
                                                          int64_t *__restrict__ seq_idx, int64_t *__restrict__ pos,
                                                          int8_t *__restrict__ strand, int64_t *__restrict__ motif_first,
                                                          unsigned long long *__restrict__ region_counts) {
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

// blk2reg[b] = region that holds position 64*b (part of the extraction stage, next to pack_kernel)
__global__ void __launch_bounds__(256) blk2reg_kernel(const int64_t *__restrict__ offsets, int64_t R, int64_t n_blocks,
                                                      int32_t *__restrict__ blk2reg) {
    const int64_t b = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    blk2reg[b] = (int32_t) find_region_bsearch(offsets, R, b * 64);
}

// ---------------------------------------------------------------- compact copy-out --

// coord = seq_idx << 32 | pos << 1 | (strand - 1): 8 bytes per hit on the host link instead of 17 (ms_result_hits_packed_host).
// bad[0] is set if a hit does not fit the format.
__global__ void __launch_bounds__(256) pack_hits_kernel(int64_t n, const int64_t *__restrict__ seq_idx, const int64_t *__restrict__ pos,
                                                        const int8_t *__restrict__ strand, uint64_t *__restrict__ coord,
                                                        unsigned int *__restrict__ bad) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t sq = (uint64_t) seq_idx[i], ps = (uint64_t) pos[i];
    if ((sq >> 32) != 0 || (ps >> 31) != 0) *bad = 1u;
    coord[i] = (sq << 32) | ((ps & 0x7FFFFFFFull) << 1) | (uint64_t) (strand[i] == 2 ? 1 : 0);
}

int launch_pack_hits(int64_t n, const int64_t *seq_idx, const int64_t *pos, const int8_t *strand, uint64_t *coord, unsigned int *bad,
                     hipStream_t st) {
    if (n == 0) return MS_OK;
    hipLaunchKernelGGL(pack_hits_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, n, seq_idx, pos, strand, coord, bad);
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

int launch_blk2reg(const int64_t *offsets, int64_t R, int64_t n_bases, int32_t *blk2reg, hipStream_t st) {
    const int64_t n_blocks = (n_bases + 63) / 64 + 1;
    hipLaunchKernelGGL(blk2reg_kernel, dim3((unsigned) ((n_blocks + 255) / 256)), dim3(256), 0, st, offsets, R, n_blocks,
                       blk2reg);
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

static PfKernel pf_kernel_for(int variant, bool meas, int *threads) {
    switch (variant) {                                                    // A/B switch MS_PF_VARIANT (MS_MEASURE=1)
        case 0: *threads = 1024; return prefilter_kernel<1024, 0, 4>;     // two groups per trip, compiler-ordered reads
        case 1: *threads = 1024; return prefilter_kernel<1024, 1, 4>;     // 2-4 groups per trip, compiler-ordered reads
        case 3: *threads = 1024; return prefilter_kernel<1024, 3, 4>;     // hand-issued reads, one full wait
        case 5: *threads = 1024; return prefilter_kernel<1024, 0, 8>;     // <= 64 VGPRs: two blocks per CU
        case 8: *threads = 768; return prefilter_kernel<768, 4, 3>;       // default form with 12 waves per CU
        case 16: *threads = 1024;                                          // engine 1 (int8 one-hot product on the matrix cores), 16 waves per CU
            return meas ? prefilter_mfma_kernel<1024, 1, 1, true> : prefilter_mfma_kernel<1024, 1, 1, false>;
        case 17: *threads = 512; return prefilter_mfma_kernel<512, 1, 1, true>;    // engine 1, 8 waves per block
        case 18: *threads = 1024; return prefilter_mfma_kernel<1024, 0, 1, true>;  // A/B: one row tile in flight per wave
        case 19: *threads = 1024; return prefilter_mfma_kernel<1024, 3, 1, true>;  // A/B: per-half maxima kept for the rare path
        case 21: *threads = 768; return prefilter_mfma_kernel<768, 2, 1, true, 2, 6>;    // A/B: row tiles of <= 2 k-blocks only, 2 x 12 waves per CU (MS_PF_BLOCKS_PER_CU=2)
        case 22: *threads = 1024; return prefilter_mfma_kernel<1024, 2, 1, true, 2, 4>;  // A/B: the same code at 16 waves per CU
        case 23: *threads = 512; return prefilter_mfma_kernel<512, 2, 1, true, 2, 6>;    // A/B: <= 2 k-blocks, 3 x 8 waves per CU (MS_PF_BLOCKS_PER_CU=3)
        case 26: *threads = 640; return prefilter_mfma_kernel<640, 2, 1, true, 2, 5>;    // A/B: <= 2 k-blocks, 2 x 10 waves per CU (MS_PF_BLOCKS_PER_CU=2), <= 96 VGPRs
        case 27: *threads = 1024; return prefilter_mfma_kernel<1024, 9, 1, true>;  // A/B: one branch per pair of tiles
        case 28: *threads = 1024; return prefilter_mfma_kernel<1024, 1, 3, true>;       // engine 3 (fp6 x fp4, 16 columns per k-block), two row tiles in flight
        case 29: *threads = 1024; return prefilter_mfma_kernel<1024, 0, 3, true>;       // engine 3, one row tile in flight
        case 30: *threads = 768; return prefilter_mfma_kernel<768, 0, 3, true, 2, 6>;   // engine 3, 2 x 12 waves per CU (MS_PF_BLOCKS_PER_CU=2)
        case 33: *threads = 768; return prefilter_mfma_kernel<768, 2, 3, true, 2, 6>;   // A/B: 30 + per-half maxima kept (2 x 12 waves, 16 registers spilled)
        case 44: *threads = 512; return prefilter_mfma_kernel<512, 3, 3, true, 2, 4>;    // 31's code in two 512-thread blocks per CU (no 3-k-block class: W <= 32 is <= 2 k-blocks of 16 columns)
        case 46: *threads = 512;                                           // engine 3 as shipped: 44 with per-WAVE hand-out (no barrier in the loop)
            return meas ? prefilter_mfma_kernel<512, 3, 3, true, 2, 4, 8> : prefilter_mfma_kernel<512, 3, 3, false, 2, 4, 8>;
        case 47: *threads = 1024; return prefilter_mfma_kernel<1024, 3, 3, true, 2, 4, 8>; // 31 with per-wave hand-out
        case 31: *threads = 1024;                                          // engine 3, two row tiles in flight, per-half maxima kept for the rare path, one 1024-thread block per CU (the default until 44)
            return meas ? prefilter_mfma_kernel<1024, 3, 3, true> : prefilter_mfma_kernel<1024, 3, 3, false>;
        case 20: *threads = 1024;                                          // engine 1, records without flags (expand_kernel decodes): the default
            return meas ? prefilter_mfma_kernel<1024, 5, 1, true> : prefilter_mfma_kernel<1024, 5, 1, false>;
        case 24: *threads = 1024;                                          // engine 2 (Walsh form: 10 columns per k-block)
            return meas ? prefilter_mfma_kernel<1024, 1, 2, true> : prefilter_mfma_kernel<1024, 1, 2, false>;
        case 25: *threads = 1024; return prefilter_mfma_kernel<1024, 0, 2, true>;  // engine 2, one row tile in flight per wave
        default: *threads = 1024; return prefilter_kernel<1024, 4, 4>;    // hand-issued reads, counted waits (default)
    }
}

int prefilter_threads(int variant) {
    int threads = 0;
    (void) pf_kernel_for(variant, true, &threads);
    return threads;
}

int prefilter_set_lds(int variant, bool meas, size_t bytes) {
    int threads;
    MS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pf_kernel_for(variant, meas, &threads)),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
    return MS_OK;
}

int launch_prefilter(const PfArgs &A, int variant, bool meas, int blocks_per_tile, int n_tiles, size_t lds_bytes, hipStream_t st) {
    int threads;
    PfKernel k = pf_kernel_for(variant, meas, &threads);
    const int64_t n_chunks = (A.n_bases + threads - 1) / threads;
    if (blocks_per_tile > n_chunks) blocks_per_tile = (int) n_chunks;
    hipLaunchKernelGGL(k, dim3((unsigned) blocks_per_tile, (unsigned) n_tiles), dim3(threads), lds_bytes, st, A);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

// The positions whose window may hold a non-ACGT base, each with its motif-independent data (needs only the sequence: it is
// launched BEFORE the pre-filter, so that the fp64 scoring of these windows can start the moment the pre-filter is done).
int launch_nlist(const DevSeq &S, int max_w, NPos *list, unsigned long long *n_list, uint64_t list_cap, const HitOut &H, hipStream_t st) {
    if (S.n_bases == 0) return MS_OK;
    const int64_t want = ((S.n_bases + 31) / 32 + 255) / 256;
    hipLaunchKernelGGL(nlist_kernel, dim3((unsigned) (want < 4096 ? want : 4096)), dim3(256), 0, st, S.nmask, S.n_bases,
                       max_w, list, n_list, list_cap);
    MS_HIP(hipGetLastError());
    hipLaunchKernelGGL(nprep_kernel, dim3(1024), dim3(256), 0, st, S, list, n_list, list_cap, H);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_neval(const DevSeq &S, const DevPwm &Pw, const int32_t *motifs, int32_t n_motifs, int strand_mask, const NPos *list,
                 const unsigned long long *n_list, uint64_t list_cap, const HitOut &H, int n_blocks_max, hipStream_t st) {
    if (S.n_bases == 0 || n_motifs == 0) return MS_OK;
    const unsigned rows = (unsigned) ((n_motifs + kNwMotifChunk - 1) / kNwMotifChunk);
    (void) n_blocks_max;
    unsigned gx = 64;                                     // many small blocks (measured against one wave of fat blocks: profiles/r02_n_fraction.log)
    if (const char *e = measure_env("MS_NEVAL_GX")) gx = (unsigned) std::max(1, atoi(e));
    dim3 grid(gx, rows);
    hipLaunchKernelGGL(neval_kernel, grid, dim3(256), 0, st, S, Pw, motifs, n_motifs, strand_mask, list, n_list, list_cap, H);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_exact_all(const DevSeq &S, const DevPwm &Pw, const int32_t *motifs, int32_t n_motifs, int strand_mask,
                     const HitOut &H, hipStream_t st) {
    if (S.n_bases == 0 || n_motifs == 0) return MS_OK;
    for (int32_t m0 = 0; m0 < n_motifs; m0 += 32768) {               // grid.y limit
        const int32_t n = n_motifs - m0 < 32768 ? n_motifs - m0 : 32768;
        dim3 grid((unsigned) ((S.n_bases + 255) / 256), (unsigned) n);
        hipLaunchKernelGGL(exact_all_kernel, grid, dim3(256), 0, st, S, Pw, motifs + m0, strand_mask, H);
        MS_HIP(hipGetLastError());
    }
    return MS_OK;
}

int launch_rescore(const DevSeq &S, const DevPwm &Pw, const uint64_t *cand, const unsigned long long *n_cand,
                   uint64_t cand_cap, const int32_t *group_motifs, int strand_mask, const HitOut &H, int n_blocks,
                   hipStream_t st) {
    hipLaunchKernelGGL(rescore_kernel, dim3((unsigned) n_blocks), dim3(256), 0, st, S, Pw, cand, n_cand, cand_cap,
                       group_motifs, strand_mask, H);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_finalize(const uint64_t *keys, int64_t n, int gbits, int rbits, int pbits, int32_t P, const DevSeq &S, int64_t *seq_idx,
                    int64_t *pos, int8_t *strand, int64_t *motif_first, unsigned long long *region_counts,
                    hipStream_t st) {
    if (n == 0) {                                    // no hits: every per-motif offset is 0
        MS_HIP(hipMemsetAsync(motif_first, 0, ((size_t) P + 1) * sizeof(int64_t), st));
        return MS_OK;
    }
    if (pbits > 0) {
        hipLaunchKernelGGL(finalize_rp_kernel, dim3((unsigned) ((n + 1023) / 1024)), dim3(256), 0, st, keys, n, rbits, pbits, P,
                           seq_idx, pos, strand, motif_first, region_counts);
        MS_HIP(hipGetLastError());
        return MS_OK;
    }
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, keys, n, gbits, P, S,
                       seq_idx, pos, strand, motif_first, region_counts);
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
