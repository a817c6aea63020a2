// ms_tail.hip -- what follows the pre-filter ("tail" of a scan), second form:
//
//   expand_kernel            candidate records (position, table group) -> ENTRIES (motif, coordinate): the group's 16 int8 row
//                            sums are recomputed exactly (integer arithmetic: the same sign bits the matrix product gave),
//                            so the pre-filter's rare path no longer decodes flags; windows that run past their region or
//                            overlap a non-ACGT base are dropped here (the latter are scored by nlist/neval)
//   (rocPRIM radix sort)     entries ordered by (motif, region, position): KEYS ONLY, 8 bytes per entry
//   rescore_ordered_kernel   fp64 re-scoring of the entries IN ORDER, in the reference's order of operations, with the
//                            reference's hit test (cscore.c:340-390): a block's hits get their place in the final arrays from a
//                            single-pass prefix sum over the blocks (decoupled look-back), so seq_idx / pos / score / strand and
//                            the per-motif offsets are written once, already in the reference's order (cscore.c:443-471)
//   pair_counts_kernel       per motif: number of sequences with >= 1 hit (stats.py:29-31)
//
// Compared with the first form (re-score unordered -> sort (key, score) pairs -> unpack) this moves half the bytes through the
// sort, reads tables and sequence in order, drops the separate finalize pass, and takes ~40 instructions per flagged tile out of
// the matrix-core kernel.  Results are identical by construction: the same windows reach the same fp64 arithmetic.
#include "ms_device.h"

namespace ms {

// ------------------------------------------------------------------------ expand --

constexpr int kEntryStage = 2048;
struct EntryStage {
    uint64_t keys[kEntryStage];
    unsigned int n;
    unsigned long long base;
};

__device__ __forceinline__ void stage_entry(EntryStage &st, const ExpandArgs &A, uint64_t key) {
    const unsigned int i = atomicAdd(&st.n, 1u);
    if (i < (unsigned int) kEntryStage) {
        st.keys[i] = key;
    } else {                                         // stage full: straight to HBM
        const unsigned long long k = atomicAdd(A.n_entries, 1ULL);
        if (k < A.entry_cap) A.entries[k] = key;
    }
}

__device__ __forceinline__ void stage_flush(EntryStage &st, const ExpandArgs &A) {       // all threads, block-uniform
    __syncthreads();
    const unsigned int n = st.n < (unsigned int) kEntryStage ? st.n : (unsigned int) kEntryStage;
    if (threadIdx.x == 0 && n > 0) st.base = atomicAdd(A.n_entries, (unsigned long long) n);
    __syncthreads();
    const unsigned long long base = st.base;
    for (unsigned int i = threadIdx.x; i < n; i += blockDim.x)
        if (base + i < A.entry_cap) A.entries[base + i] = st.keys[i];
    __syncthreads();
    if (threadIdx.x == 0) st.n = 0;
    __syncthreads();
}

__device__ __forceinline__ int sbyte(uint32_t word, uint32_t b) { return (int) __builtin_amdgcn_sbfe((int) word, b * 8u, 8u); }

// 16 lanes per record: lane f of the group owns field f of the record's table group (field n: motif slot n >> 1, even n
// forward, odd n reverse) = row (j & 3) + 8 (j >> 2) + 4 h of the row tile, j = 15 - f (ms_internal.h, "engine 1").
__global__ void __launch_bounds__(256) expand_kernel(const ExpandArgs A) {
    __shared__ EntryStage st;
    if (threadIdx.x == 0) st.n = 0;
    __syncthreads();
    unsigned long long n = *A.n_cand;
    if (n > A.cand_cap) n = A.cand_cap;
    const uint32_t f = threadIdx.x & 15u;
    const unsigned long long per_round = (unsigned long long) gridDim.x * (blockDim.x / 16);
    const unsigned long long rounds = (n + per_round - 1) / per_round;
    for (unsigned long long rd = 0; rd < rounds; rd++) {
        const unsigned long long i = rd * per_round + (unsigned long long) blockIdx.x * (blockDim.x / 16) + (threadIdx.x >> 4);
        bool mine = false;
        uint64_t rec = 0;
        int64_t g = 0;
        uint32_t q = 0;
        if (i < n) {
            rec = A.cand[i];
            g = (int64_t) (rec >> 30);
            q = (uint32_t) (rec >> 16) & 0x3FFFu;
            const uint32_t fl = (uint32_t) rec & 0xFFFFu;
            if (fl != 0) {                                   // the producer decoded the flags itself (engines 0 / 2, A/B variants)
                mine = (fl >> f) & 1u;
            } else {
                const uint32_t rt = q >> 1, h = q & 1u, j = 15u - f;
                const uint32_t row = (j & 3u) + 8u * (j >> 2) + 4u * h;
                const int nk = A.rt_nk[rt];
                const uint4 *__restrict__ t = A.tables + A.rt_off16[rt] + row;
                const uint64_t cw = code_window(A.S.codes, g);
                int acc = 0;
                for (int kb = 0; kb < nk; kb++) {
#pragma unroll
                    for (int kh = 0; kh < 2; kh++) {
                        const uint4 v = t[kb * 64 + kh * 32];
                        const uint32_t c8 = (uint32_t) (cw >> (16 * kb + 8 * kh)) & 0xFFu;
                        acc += sbyte(v.x, c8 & 3u) + sbyte(v.y, (c8 >> 2) & 3u) + sbyte(v.z, (c8 >> 4) & 3u) + sbyte(v.w, (c8 >> 6) & 3u);
                    }
                }
                mine = acc >= 0;                             // the sign bit the matrix product produced for this row
            }
        }
        const unsigned long long bal = __ballot(mine);
        const uint32_t grp = (uint32_t) (bal >> (threadIdx.x & 48u)) & 0xFFFFu;
        if (i < n && (f & 1u) == 0 && ((grp >> f) & 3u)) {   // slot leader: forward or reverse (or both) flagged
            const int32_t pm = A.group_motifs[q * kGroupSlots + (f >> 1)];
            if (pm >= 0) {
                const int W = A.width[pm];
                const int64_t r = find_region(A.S, g);
                const uint32_t nw = n_window(A.S.nmask, g);
                if (g + W <= A.S.offsets[r + 1] && !(nw & low_mask(W))) {     // inside its region (cscore.c:340); no N (those: neval_kernel)
                    const uint64_t coord = A.pbits ? (((uint64_t) r << A.pbits) | (uint64_t) (g - A.S.offsets[r])) : (uint64_t) g;
                    stage_entry(st, A, ((uint64_t) pm << A.cbits) | coord);
                }
            }
        }
        __syncthreads();
        const bool full = st.n > (unsigned int) (kEntryStage - 256);
        __syncthreads();
        if (full) stage_flush(st, A);
    }
    stage_flush(st, A);
}

// --------------------------------------------------------------- ordered re-scoring --

constexpr int kRPerThread = 4;
constexpr int kRTile = 256 * kRPerThread;
constexpr unsigned long long kStMask = (1ULL << 62) - 1ULL;

// state word of a tile: [63:62] 0 = nothing yet, 1 = the tile's own hit count, 2 = hit count of all tiles up to and including
// it; [61:0] the count.  One naturally aligned 8-byte word written by ONE agent-scope store and polled with agent-scope loads
// (MI355X_MICROARCH.md "granule": needs no other ordering).
__device__ __forceinline__ void st_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long st_load(unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(256) rescore_ordered_kernel(const OrderedArgs A) {
    __shared__ unsigned int s_tile;
    __shared__ unsigned int s_wave[4];
    __shared__ unsigned long long s_base;
    if (threadIdx.x == 0) s_tile = atomicAdd(A.tile_counter, 1u);      // tiles are handed out in order: every earlier tile is already running
    __syncthreads();
    const uint64_t tile = s_tile;
    const uint64_t e0 = tile * kRTile + (uint64_t) threadIdx.x * kRPerThread;
    const uint64_t cmask = (1ULL << A.cbits) - 1ULL, pmask = (1ULL << A.pbits) - 1ULL;

    int64_t reg[kRPerThread], ps[kRPerThread];
    double sf[kRPerThread], sr[kRPerThread];
    uint32_t hit[kRPerThread], motif[kRPerThread];
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < kRPerThread; j++) {
        hit[j] = 0;
        motif[j] = 0;
        reg[j] = ps[j] = 0;
        sf[j] = sr[j] = 0.0;
        if (e0 + j < A.n) {
            const uint64_t k = A.keys[e0 + j];
            const uint32_t m = (uint32_t) (k >> A.cbits);
            const uint64_t coord = k & cmask;
            int64_t g, r;
            if (A.pbits) { r = (int64_t) (coord >> A.pbits); g = A.S.offsets[r] + (int64_t) (coord & pmask); }
            else { g = (int64_t) coord; r = find_region(A.S, g); }
            motif[j] = m;
            reg[j] = r;
            ps[j] = g - A.S.offsets[r];
            const int W = A.Pw.width[m];
            double fwd, rev;
            if (W <= kMaxFastWidth) score_window32(A.Pw.tab2 + A.Pw.tab_off[m], W, code_window(A.S.codes, g), n_window(A.S.nmask, g), fwd, rev);
            else score_window(A.S, A.Pw.tab2 + A.Pw.tab_off[m], W, g, fwd, rev);
            // the reference's normalisation and threshold test (cscore.c:356-358 / 373-375); the divides are only paid above
            // a proven floor of the raw sum (ms_api.hip, raw_floor)
            const double floor_ = A.Pw.raw_floor[m];
            const bool try_f = (A.strand_mask & 1) && !(fwd < floor_), try_r = (A.strand_mask & 2) && !(rev < floor_);
            if (try_f || try_r) {
                const double max_raw = A.Pw.max_raw[m], cutoff = A.Pw.cutoff[m];
                if (try_f) { sf[j] = fwd / max_raw; if (sf[j] - cutoff >= -1e-10) hit[j] |= 1u; }
                if (try_r) { sr[j] = rev / max_raw; if (sr[j] - cutoff >= -1e-10) hit[j] |= 2u; }
            }
            cnt += (hit[j] & 1u) + (hit[j] >> 1);
        }
    }
    // exclusive prefix of the hit counts inside the block
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if (lane >= (uint32_t) o) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, agg = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { if ((uint32_t) w < wave) before += s_wave[w]; agg += s_wave[w]; }
    // where the block's hits start: sum over all earlier tiles (single-pass, decoupled look-back)
    if (threadIdx.x == 0) {
        unsigned long long excl = 0;
        if (tile > 0) {
            st_store(A.tile_state + tile, (1ULL << 62) | agg);
            for (int64_t i = (int64_t) tile - 1; i >= 0; i--) {
                unsigned long long s = st_load(A.tile_state + i);
                for (unsigned int spin = 0; (s >> 62) == 0 && spin < (1u << 24); spin++) { __builtin_amdgcn_s_sleep(2); s = st_load(A.tile_state + i); }
                if ((s >> 62) == 0) { *A.error = 1u; break; }                  // bounded: report instead of hanging
                excl += s & kStMask;
                if ((s >> 62) == 2) break;
            }
        }
        st_store(A.tile_state + tile, (2ULL << 62) | (excl + agg));
        s_base = excl;
    }
    __syncthreads();
    uint64_t o = s_base + before + (incl - cnt);
    // per-motif offsets: every motif after the previous entry's up to this entry's starts at this entry's first output slot
    uint32_t prev_m = 0xFFFFFFFFu;
    if (e0 > 0 && e0 <= A.n) prev_m = (uint32_t) (A.keys[e0 - 1] >> A.cbits);
#pragma unroll
    for (int j = 0; j < kRPerThread; j++) {
        if (e0 + j < A.n) {
            if (prev_m != motif[j])
                for (int64_t q = prev_m == 0xFFFFFFFFu ? 0 : (int64_t) prev_m + 1; q <= (int64_t) motif[j]; q++) A.motif_first[q] = (int64_t) o;
            prev_m = motif[j];
            if (hit[j] & 1u) {
                if (o < A.cap) { A.seq_idx[o] = reg[j]; A.pos[o] = ps[j]; A.score[o] = sf[j]; A.strand[o] = 1; }
                o++;
            }
            if (hit[j] & 2u) {                                                 // '+' before '-' at the same position (cscore.c:356-389)
                if (o < A.cap) { A.seq_idx[o] = reg[j]; A.pos[o] = ps[j]; A.score[o] = sr[j]; A.strand[o] = 2; }
                o++;
            }
            if (e0 + j == A.n - 1) {                                           // the last entry: total, and the motifs behind it are empty
                for (int64_t q = (int64_t) motif[j] + 1; q <= A.P; q++) A.motif_first[q] = (int64_t) o;
                *A.n_hits = o;
            }
        }
    }
}

// ------------------------------------------------------------------- region counts --

// per motif: number of sequences with >= 1 hit (stats.py:29-31): a hit opens a new (motif, sequence) pair if it is the first of its
// motif or its sequence differs from the previous hit's.  One atomic per (wave, motif).
__global__ void __launch_bounds__(256) pair_counts_kernel(const unsigned long long *__restrict__ n_hits, uint64_t cap,
                                                          const int64_t *__restrict__ motif_first, int32_t P,
                                                          const int64_t *__restrict__ seq_idx,
                                                          unsigned long long *__restrict__ region_counts) {
    unsigned long long n = *n_hits;
    if (n > cap) n = cap;
    const unsigned long long stride = (unsigned long long) gridDim.x * blockDim.x;
    const unsigned long long rounds = (n + stride - 1) / stride;
    for (unsigned long long rd = 0; rd < rounds; rd++) {
        const unsigned long long i = rd * stride + (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
        const bool live = i < n;
        int32_t m = -1;
        bool opens = false;
        if (live) {
            int32_t lo = 0, hi = P;                               // motif_first[lo] <= i < motif_first[hi]
            while (hi - lo > 1) {
                const int32_t mid = (lo + hi) >> 1;
                if ((unsigned long long) motif_first[mid] <= i) lo = mid; else hi = mid;
            }
            m = lo;
            opens = (unsigned long long) motif_first[m] == i || seq_idx[i] != seq_idx[i - 1];
        }
        unsigned long long todo = __ballot(live && opens);
        while (todo) {
            const int leader = __ffsll((long long) todo) - 1;
            const int32_t mm = __shfl(m, leader);
            const unsigned long long same = __ballot(live && opens && m == mm);
            if ((int) (threadIdx.x & 63) == leader) atomicAdd(&region_counts[mm], (unsigned long long) __popcll(same));
            todo &= ~same;
        }
    }
}

// ---------------------------------------------------------------------- launchers --

int launch_expand(const ExpandArgs &A, int n_blocks, hipStream_t st) {
    hipLaunchKernelGGL(expand_kernel, dim3((unsigned) n_blocks), dim3(256), 0, st, A);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

size_t ordered_tiles(uint64_t n_entries) { return (size_t) ((n_entries + kRTile - 1) / kRTile); }

int launch_rescore_ordered(const OrderedArgs &A, hipStream_t st) {
    if (A.n == 0) return MS_OK;
    hipLaunchKernelGGL(rescore_ordered_kernel, dim3((unsigned) ordered_tiles(A.n)), dim3(256), 0, st, A);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_pair_counts(const unsigned long long *n_hits, uint64_t cap, const int64_t *motif_first, int32_t P, const int64_t *seq_idx,
                       unsigned long long *region_counts, int n_blocks, hipStream_t st) {
    hipLaunchKernelGGL(pair_counts_kernel, dim3((unsigned) n_blocks), dim3(256), 0, st, n_hits, cap, motif_first, P, seq_idx, region_counts);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

}  // namespace ms
