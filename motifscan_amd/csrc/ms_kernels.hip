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
//   prefilter_kernel  16-bit integer upper bound of both strand scores for EVERY window, PWM
//                     2-mer tables in LDS, one lane per window start; emits candidates
//   nwindow_kernel    fp64 scoring of the windows that overlap a non-ACGT base
//   exact_all_kernel  fp64 scoring of every window for motifs the pre-filter cannot take
//   rescore_kernel    fp64 scoring of the candidates, in the reference's order of operations,
//                     and the reference's hit test (cscore.c:356-358, 373-375)
//   finalize_kernel   sorted keys -> (seq_idx, pos, strand), per-motif offsets, region counts
//   score_kernel      c_score: first W bases of every sequence    (cscore.c:191-224)
#include "ms_kernels.h"

namespace ms {

// ------------------------------------------------------------------------ helpers --

__device__ __forceinline__ uint64_t code_window(const uint32_t *__restrict__ codes, int64_t g) {
    const int64_t wi = g >> 4;
    const uint32_t sh = ((uint32_t) g & 15u) * 2u;
    const uint32_t w0 = codes[wi], w1 = codes[wi + 1], w2 = codes[wi + 2];
    const uint64_t lo = ((uint64_t) w1 << 32) | w0;
    return sh ? (lo >> sh) | ((uint64_t) w2 << (64u - sh)) : lo;
}

__device__ __forceinline__ uint32_t n_window(const uint32_t *__restrict__ nmask, int64_t g) {
    const int64_t wi = g >> 5;
    const uint32_t sh = (uint32_t) g & 31u;
    const uint32_t w0 = nmask[wi], w1 = nmask[wi + 1];
    return sh ? (w0 >> sh) | (w1 << (32u - sh)) : w0;
}

__device__ __forceinline__ uint32_t low_mask(int w) { return w >= 32 ? 0xFFFFFFFFu : ((1u << w) - 1u); }

// region r with offsets[r] <= g < offsets[r+1]  (empty regions are skipped by construction)
__device__ __forceinline__ int64_t find_region(const int64_t *__restrict__ offsets, int64_t R, int64_t g) {
    int64_t lo = 0, hi = R;          // invariant: offsets[lo] <= g, answer in [lo, hi)
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

// fp64 scores of one window in the reference's order: c = 0..W-1, forward adds M[row][c],
// reverse adds M[3-row][W-1-c], non-ACGT adds nothing (cscore.c:345-354).
__device__ __forceinline__ void score_window(const DevSeq &S, const double2 *__restrict__ tab, int W,
                                             int64_t g, double &fwd, double &rev) {
    fwd = 0.0;
    rev = 0.0;
    for (int c0 = 0; c0 < W; c0 += 32) {
        const uint64_t cw = code_window(S.codes, g + c0);
        const uint32_t nw = n_window(S.nmask, g + c0);
        const int n = (W - c0) < 32 ? (W - c0) : 32;
        for (int c = 0; c < n; c++) {
            if ((nw >> c) & 1u) continue;
            const uint32_t b = (uint32_t) (cw >> (2 * c)) & 3u;
            const double2 t = tab[(c0 + c) * 4 + b];
            fwd += t.x;
            rev += t.y;
        }
    }
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
    const double max_raw = Pw.max_raw[motif];
    const double cutoff = Pw.cutoff[motif];
    if (strand_mask & 1) {
        const double s = fwd / max_raw;
        if (s - cutoff >= -1e-10) emit_hit(H, motif, g, 0u, s);
    }
    if (strand_mask & 2) {
        const double s = rev / max_raw;
        if (s - cutoff >= -1e-10) emit_hit(H, motif, g, 1u, s);
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
// pair of quads.  One global atomic per find would put every wave of the chip on ONE address
// (measured: the whole kernel then runs at the ~90 M atomics/s a single word sustains).  So each
// wave appends to its own queue in LDS with ballot/mbcnt ranks (no atomics at all) and spills it
// to the global list with a single atomicAdd per >= 64 entries.

__device__ __forceinline__ void wq_flush(uint64_t *__restrict__ wbuf, uint32_t n, uint64_t *__restrict__ cand,
                                      unsigned long long *__restrict__ n_cand, uint64_t cand_cap) {
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(n_cand, (unsigned long long) n);
    base = __shfl(base, 0);
    for (uint32_t i = lane; i < n; i += 64)
        if (base + i < cand_cap) cand[base + i] = wbuf[i];
}

// Entered by the WHOLE wave (uniform branch) when any lane has a flagged field in this quad.
// Returns the new queue length.
__device__ __noinline__ uint32_t emit_candidates(uint64_t *__restrict__ wbuf, uint32_t n, uint64_t *__restrict__ cand,
                                                 unsigned long long *__restrict__ n_cand, uint64_t cand_cap, uint4 acc,
                                                 int32_t quad, int64_t g, bool live) {
    const uint32_t w[4] = {acc.x & 0x80008000u, acc.y & 0x80008000u, acc.z & 0x80008000u, acc.w & 0x80008000u};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const bool flagged = live && w[k] != 0;
        const unsigned long long mask = __ballot(flagged);
        if (mask == 0) continue;
        const uint32_t n_new = (uint32_t) __popcll(mask);
        if (n + n_new > (uint32_t) kWqCap) {
            wq_flush(wbuf, n, cand, n_cand, cand_cap);
            n = 0;
        }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
        if (flagged) {
            const uint32_t strands = ((w[k] & 0x8000u) ? 1u : 0u) | ((w[k] & 0x80000000u) ? 2u : 0u);
            wbuf[n + rank] = cand_pack((uint64_t) g, (uint32_t) (quad * 4 + k), strands);
        }
        n += n_new;
    }
    return n;
}

struct PfWave {
    uint64_t *wbuf;      // this wave's queue in LDS (kWqCap entries)
    uint32_t n;          // entries queued (wave-uniform)
    int64_t g;           // this lane's window start
    bool live;           // g < n_bases
};

#define MS_PF_EMIT(ACC, QUAD)                                                                               \
    W.n = __builtin_amdgcn_readfirstlane(                                                                   \
        emit_candidates(W.wbuf, W.n, A.cand, A.n_cand, A.cand_cap, ACC, QUAD, W.g, W.live))

// All quads of one class (same group count G): per quad G LDS reads of 16 bytes (four motifs x
// {fwd,rev} 16-bit fields) and (G-1) x 4 packed adds.  `code16[g]` is the lane's 2-mer code at
// group g (0..15); a table row of one (quad, group) is 16 codes x 16 B = 256 B = every LDS bank
// exactly once, so the read is conflict-free whatever the codes are.
template <int G>
__device__ __forceinline__ void prefilter_class(const PfArgs &A, const uint4 *__restrict__ lds4, uint32_t base16,
                                                int n_quads, int32_t first_quad, const uint32_t (&code16)[kMaxGroups],
                                                PfWave &W) {
    uint32_t a[G];
#pragma unroll
    for (int k = 0; k < G; k++) a[k] = base16 + (uint32_t) k * 16u + code16[k];
    int q = 0;
    for (; q + 2 <= n_quads; q += 2) {
        uint4 acc0 = lds4[a[0]];
        uint4 acc1 = lds4[a[0] + G * 16];
#pragma unroll
        for (int k = 1; k < G; k++) {
            add4(acc0, lds4[a[k]]);
            add4(acc1, lds4[a[k] + G * 16]);
        }
        const uint32_t any = (acc0.x | acc0.y | acc0.z | acc0.w | acc1.x | acc1.y | acc1.z | acc1.w) & 0x80008000u;
        if (__any(any != 0)) {
            MS_PF_EMIT(acc0, first_quad + q);
            MS_PF_EMIT(acc1, first_quad + q + 1);
        }
#pragma unroll
        for (int k = 0; k < G; k++) a[k] += 2 * G * 16;
    }
    if (q < n_quads) {
        uint4 acc0 = lds4[a[0]];
#pragma unroll
        for (int k = 1; k < G; k++) add4(acc0, lds4[a[k]]);
        const uint32_t any = (acc0.x | acc0.y | acc0.z | acc0.w) & 0x80008000u;
        if (__any(any != 0)) MS_PF_EMIT(acc0, first_quad + q);
    }
}

#define MS_PF_CASE(GG)                                                                      \
    case GG:                                                                                \
        prefilter_class<GG>(A, lds4, base16, nq, first_quad, code16, W);                    \
        break;

// grid = (blocks per tile, tiles).  One block per CU (the tile's tables fill the LDS), 16 waves,
// each wave takes 64 consecutive window starts per iteration.  Dynamic LDS = tables of the
// largest tile, then 16 wave queues of kWqCap candidates.
__global__ void __launch_bounds__(kPfThreads) prefilter_kernel(const PfArgs A) {
    extern __shared__ uint4 lds4[];
    const TileDesc *__restrict__ T = A.tiles + blockIdx.y;
    const uint32_t len16 = T->table_len16;
    const uint4 *__restrict__ src = A.tables + T->table_off16;
    for (uint32_t i = threadIdx.x; i < len16; i += kPfThreads) lds4[i] = src[i];
    __syncthreads();
    const int n_classes = T->n_classes;
    const int32_t tile_first_quad = T->first_quad;
    PfWave W;
    W.wbuf = reinterpret_cast<uint64_t *>(lds4 + A.wq_off16) + (threadIdx.x >> 6) * kWqCap;
    W.n = 0;

    for (int64_t chunk = blockIdx.x; chunk < A.n_chunks; chunk += gridDim.x) {
        W.g = chunk * kPfThreads + threadIdx.x;
        W.live = W.g < A.n_bases;
        const uint64_t cw = code_window(A.codes, W.live ? W.g : 0);
        uint32_t code16[kMaxGroups];
#pragma unroll
        for (int k = 0; k < kMaxGroups; k++) code16[k] = (uint32_t) (cw >> (4 * k)) & 15u;

        uint32_t base16 = 0;
        int32_t first_quad = tile_first_quad;
        for (int c = 0; c < n_classes; c++) {
            const int G = T->cls[c].G;
            const int nq = T->cls[c].n_quads;
            switch (G) {
                MS_PF_CASE(1) MS_PF_CASE(2) MS_PF_CASE(3) MS_PF_CASE(4)
                MS_PF_CASE(5) MS_PF_CASE(6) MS_PF_CASE(7) MS_PF_CASE(8)
                MS_PF_CASE(9) MS_PF_CASE(10) MS_PF_CASE(11) MS_PF_CASE(12)
                MS_PF_CASE(13) MS_PF_CASE(14) MS_PF_CASE(15) MS_PF_CASE(16)
                default: break;
            }
            base16 += (uint32_t) (G * 16 * nq);
            first_quad += nq;
        }
    }
    if (W.n > 0) wq_flush(W.wbuf, W.n, A.cand, A.n_cand, A.cand_cap);
}

// -------------------------------------------------------------------- fp64 kernels --

// grid = (ceil(n_bases/256), ceil(n_fast/kNwMotifChunk)).  Almost every thread leaves at the
// first test: only windows that overlap a non-ACGT base are scored here (the pre-filter packs
// such bases as 'A', so its answer for these windows means nothing and rescore_kernel skips them).
__global__ void __launch_bounds__(256) nwindow_kernel(const DevSeq S, const DevPwm Pw, const int32_t *__restrict__ motifs,
                                                      int32_t n_motifs, int strand_mask, const HitOut H) {
    const int64_t g = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= S.n_bases) return;
    const uint32_t nw = n_window(S.nmask, g);
    if (nw == 0) return;
    const int64_t r = find_region(S.offsets, S.R, g);
    const int64_t end = S.offsets[r + 1];
    const int m0 = blockIdx.y * kNwMotifChunk;
    const int m1 = min(m0 + kNwMotifChunk, n_motifs);
    for (int m = m0; m < m1; m++) {
        const int32_t p = motifs[m];
        const int W = Pw.width[p];
        if ((nw & low_mask(W)) == 0) continue;
        if (g + W > end) continue;
        double fwd, rev;
        score_window(S, Pw.tab2 + Pw.tab_off[p], W, g, fwd, rev);
        test_and_emit(H, Pw, (uint32_t) p, g, fwd, rev, strand_mask);
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
    const int64_t r = find_region(S.offsets, S.R, g);
    if (g + W > S.offsets[r + 1]) return;
    double fwd, rev;
    score_window(S, Pw.tab2 + Pw.tab_off[p], W, g, fwd, rev);
    test_and_emit(H, Pw, (uint32_t) p, g, fwd, rev, strand_mask);
}

__global__ void __launch_bounds__(256) rescore_kernel(const DevSeq S, const DevPwm Pw, const uint64_t *__restrict__ cand,
                                                      const unsigned long long *__restrict__ n_cand, uint64_t cand_cap,
                                                      const int32_t *__restrict__ quad_motifs, int strand_mask,
                                                      const HitOut H) {
    unsigned long long n = *n_cand;
    if (n > cand_cap) n = cand_cap;
    for (unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long) gridDim.x * blockDim.x) {
        const uint64_t c = cand[i];
        const int32_t pm = quad_motifs[(uint32_t) (c >> 2) & 0xFFFFu];       // candidate carries its table slot
        if (pm < 0) continue;
        const uint32_t p = (uint32_t) pm;
        const int64_t g = (int64_t) (c >> 18);
        const int W = Pw.width[p];
        const int64_t r = find_region(S.offsets, S.R, g);
        if (g + W > S.offsets[r + 1]) continue;                        // window runs past its region (cscore.c:340)
        if (n_window(S.nmask, g) & low_mask(W)) continue;              // scored by nwindow_kernel
        double fwd, rev;
        score_window(S, Pw.tab2 + Pw.tab_off[p], W, g, fwd, rev);
        test_and_emit(H, Pw, p, g, fwd, rev, strand_mask);
    }
}

// ----------------------------------------------------------------------- finalize --

__global__ void __launch_bounds__(256) finalize_kernel(const uint64_t *__restrict__ keys, int64_t n, int gbits,
                                                       const int64_t *__restrict__ offsets, int64_t R,
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
        const int64_t r = find_region(offsets, R, g);
        seq_idx[i] = r;
        pos[i] = g - offsets[r];
        strand[i] = (int8_t) ((k & 1ULL) ? 2 : 1);
        bool first_of_motif = (i == 0);
        new_pair = true;
        if (i > 0) {
            const uint64_t kp = keys[i - 1];
            const uint32_t mp = (uint32_t) (kp >> (gbits + 1));
            first_of_motif = mp != motif;
            if (!first_of_motif) {
                const int64_t gp = (int64_t) ((kp >> 1) & gmask);
                new_pair = gp < offsets[r];            // previous hit of this motif lies in an earlier region
            }
        }
        if (first_of_motif) motif_first[motif] = i;
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

// ---------------------------------------------------------------------- launchers --

int launch_pack(const uint8_t *ascii, int64_t n_bases, uint32_t *codes, uint32_t *nmask, hipStream_t st) {
    const int64_t n_units = (n_bases + 31) / 32;
    if (n_units == 0) return MS_OK;
    const int aligned16 = (reinterpret_cast<uintptr_t>(ascii) & 15u) == 0;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned) ((n_units + 255) / 256)), dim3(256), 0, st, ascii, n_bases,
                       codes, nmask, n_units, aligned16);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int prefilter_set_lds(size_t bytes) {
    MS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(prefilter_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
    return MS_OK;
}

int launch_prefilter(const PfArgs &A, int blocks_per_tile, int n_tiles, size_t lds_bytes, hipStream_t st) {
    hipLaunchKernelGGL(prefilter_kernel, dim3((unsigned) blocks_per_tile, (unsigned) n_tiles), dim3(kPfThreads),
                       lds_bytes, st, A);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_nwindow(const DevSeq &S, const DevPwm &Pw, const int32_t *motifs, int32_t n_motifs, int strand_mask,
                   const HitOut &H, hipStream_t st) {
    if (S.n_bases == 0 || n_motifs == 0) return MS_OK;
    dim3 grid((unsigned) ((S.n_bases + 255) / 256), (unsigned) ((n_motifs + kNwMotifChunk - 1) / kNwMotifChunk));
    hipLaunchKernelGGL(nwindow_kernel, grid, dim3(256), 0, st, S, Pw, motifs, n_motifs, strand_mask, H);
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
                   uint64_t cand_cap, const int32_t *quad_motifs, int strand_mask, const HitOut &H, int n_blocks,
                   hipStream_t st) {
    hipLaunchKernelGGL(rescore_kernel, dim3((unsigned) n_blocks), dim3(256), 0, st, S, Pw, cand, n_cand, cand_cap,
                       quad_motifs, strand_mask, H);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_finalize(const uint64_t *keys, int64_t n, int gbits, const int64_t *offsets, int64_t R, int64_t *seq_idx,
                    int64_t *pos, int8_t *strand, int64_t *motif_first, unsigned long long *region_counts,
                    hipStream_t st) {
    if (n == 0) return MS_OK;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, keys, n, gbits, offsets,
                       R, seq_idx, pos, strand, motif_first, region_counts);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_score(const DevSeq &S, const DevPwm &Pw, int strand_mask, double *out, hipStream_t st) {
    if (S.R == 0 || Pw.P == 0) return MS_OK;
    for (int32_t p0 = 0; p0 < Pw.P; p0 += 32768) {
        const int32_t n = Pw.P - p0 < 32768 ? Pw.P - p0 : 32768;
        DevPwm sub = Pw;
        sub.tab_off += p0; sub.width += p0; sub.max_raw += p0; sub.cutoff += p0; sub.P = n;
        dim3 grid((unsigned) ((S.R + 255) / 256), (unsigned) n);
        hipLaunchKernelGGL(score_kernel, grid, dim3(256), 0, st, S, sub, strand_mask, out + (int64_t) p0 * S.R);
        MS_HIP(hipGetLastError());
    }
    return MS_OK;
}

}  // namespace ms
