// ms_order.hip -- the ORDERED tail of a scan: fp64 re-scoring of the candidates in position order, and the placement of every hit
// at its final rank -- no sort.
//
// The reference returns a motif's hits in (sequence, position, '+' before '-') order (cscore.c:336-390), motif after motif
// (cscore.c:443-471).  Rounds 1-2 re-scored the candidates in arbitrary order and ordered the 6e7 (key, score) pairs of a full
// configs[3] set with a five-pass radix sort (rocPRIM: 2.9 ms of a 26 ms scan, 10 GB of traffic).  Here the order falls out of how
// the work is laid out:
//   * the pre-filter hands its window starts out in UNITS of consecutive positions and writes each unit's candidate records into
//     the unit's own slots of the list (ms_kernels.hip, emit_rec: no atomics);
//   * a CHUNK = a fixed number k of consecutive units = a contiguous range of positions.  One block re-scores a chunk's records,
//     stages the hits in LDS, groups them by motif (counting sort: LDS histogram + exclusive scan) and ranks each hit inside its
//     (chunk, motif) group -- a handful of hits -- by key.  The chunk's hits leave the block sorted by (motif, coordinate, strand),
//     to one contiguous piece of the hit list, together with the row cnt[chunk][motif] of a count matrix;
//   * since chunk c holds only positions below chunk c + 1's, the final rank of a hit is
//         motif_first[m] + sum_{c' < c} cnt[c'][m] + (its rank inside its (chunk, motif) group):
//     a column-wise prefix sum over the count matrix (small kernels over n_chunks x P 16-bit cells) and one placement kernel
//     that reads the chunk-sorted hits once and writes the result arrays (seq_idx, pos, score, strand) once.
// Traffic per hit: 16 B written + 16 B read + 25 B written (+ ~10 B of matrix) against 16 + 5 x 32 + 8 + 17 B before.
//
// Limits (the host keeps the sorted tail for the rest, scan_locked): P <= kOrdMaxMotifs (LDS histogram), no motif on the all-fp64
// path (exact_all_kernel emits unordered), hit coordinates of the (region, position) form, a chunk's hits <= kOrdStage (else the
// host halves the chunk and runs the pass again), a unit's records <= its slots (else the slots grow and the pass runs again).
#include <algorithm>

#include "ms_device.h"

namespace ms {

namespace {

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total) {        // exclusive prefix over the 64 lanes
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o);
        if ((int) (threadIdx.x & 63) >= o) x += y;
    }
    total = __shfl(x, 63);
    return x - v;
}

// exclusive prefix sum of a[0..n) in LDS, in place; returns the total.  256 threads (all of them call); scratch: 8 words of LDS.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t *a, int n, uint32_t *scratch) {
    const int tid = threadIdx.x, per = (n + 255) / 256;
    uint32_t local = 0;
    for (int i = tid * per; i < n && i < (tid + 1) * per; i++) local += a[i];
    uint32_t wtot;
    const uint32_t pre = wave_excl_scan(local, wtot);
    __syncthreads();                                                   // (scratch may still be read by the previous call's callers)
    if ((tid & 63) == 63) scratch[tid >> 6] = wtot;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
    for (int w = 0; w < 4; w++) { if (w < (tid >> 6)) wbase += scratch[w]; total += scratch[w]; }
    uint32_t run = pre + wbase;
    for (int i = tid * per; i < n && i < (tid + 1) * per; i++) { const uint32_t v = a[i]; a[i] = run; run += v; }
    __syncthreads();
    return total;
}

}  // namespace

// grid: blocks loop over chunks.  Dynamic LDS: stage keys [kOrdStage] u64 | stage vals [kOrdStage] f64 | hist [P] u32 | start [P] u32 |
// cursor [P] u32 | src_off [n_src + 1] u32 | order [kOrdStage] u16
__global__ void __launch_bounds__(256) rescore_ordered_kernel(const OrdArgs O) {
    extern __shared__ unsigned char ord_lds[];
    uint64_t *st_key = reinterpret_cast<uint64_t *>(ord_lds);
    double *st_val = reinterpret_cast<double *>(st_key + kOrdStage);
    uint32_t *hist = reinterpret_cast<uint32_t *>(st_val + kOrdStage);
    uint32_t *start = hist + O.P;
    uint32_t *cursor = start + O.P;
    const int n_src = O.n_tiles * O.k;                                 // (tile, unit of the chunk) sources of records
    uint32_t *src_off = cursor + O.P;
    uint16_t *order = reinterpret_cast<uint16_t *>(src_off + n_src + 1);
    __shared__ uint32_t s_n, s_scratch[8];
    __shared__ unsigned long long s_base;
    const DevSeq &S = O.S;
    const DevPwm &Pw = O.Pw;
    const int tid = threadIdx.x;
    const bool both = O.strand_mask == 3;
    for (int64_t c = blockIdx.x; c < O.n_chunks; c += gridDim.x) {
        for (int m = tid; m < O.P; m += 256) { hist[m] = 0; cursor[m] = 0; }
        if (tid == 0) s_n = 0;
        for (int s = tid; s <= n_src; s += 256) {
            uint32_t cntv = 0;
            if (s < n_src) {
                const int64_t u = c * O.k + (s % O.k);
                if (u < O.n_units) {
                    cntv = O.unit_cnt[(int64_t) (s / O.k) * O.n_units + u];
                    if (cntv > O.unit_slots) {                         // the unit dropped records: the host grows the slots and runs the pass again
                        atomicOr(O.overflow, 1u);
                        atomicMax(O.overflow + 1, cntv);
                        cntv = O.unit_slots;
                    }
                }
            }
            src_off[s] = cntv;
        }
        __syncthreads();
        const uint32_t total_rec = block_excl_scan(src_off, n_src + 1, s_scratch);       // src_off[s] = first record of source s; src_off[n_src] = total
        if (tid == 0 && total_rec) atomicAdd(O.n_rec, (unsigned long long) total_rec);
        // ---- re-score the chunk's records: kOrdU of them in flight per thread (the kernel is a chain of dependent gathers)
        for (uint32_t base = 0; base < total_rec; base += 256u * kOrdU) {
            uint64_t rec[kOrdU];
            bool live[kOrdU];
#pragma unroll
            for (int u = 0; u < kOrdU; u++) {
                const uint32_t idx = base + (uint32_t) u * 256u + (uint32_t) tid;
                live[u] = idx < total_rec;
                rec[u] = 0;
                if (live[u]) {
                    int lo = 0, hi = n_src;                                   // source s with src_off[s] <= idx < src_off[s + 1]
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (src_off[mid] <= idx) lo = mid; else hi = mid; }
                    const int64_t unit = c * O.k + (lo % O.k);
                    rec[u] = O.cand[((int64_t) (lo / O.k) * O.n_units + unit) * O.unit_slots + (idx - src_off[lo])];
                }
            }
            int64_t g[kOrdU], lo_r[kOrdU];
            uint64_t cw[kOrdU];
            uint32_t nw[kOrdU], flags[kOrdU];
            int32_t group[kOrdU], pm[kOrdU];
#pragma unroll
            for (int u = 0; u < kOrdU; u++) {
                g[u] = (int64_t) (rec[u] >> 30);
                group[u] = (int32_t) ((rec[u] >> 16) & 0x3FFFu);
                const uint32_t f = (uint32_t) rec[u] & 0xFFFFu;
                flags[u] = both ? (f | (f >> 1)) & 0x5555u : f;          // a motif's two strands are re-scored together anyway
                lo_r[u] = S.blk2reg[g[u] >> 6];
                cw[u] = code_window(S.codes, g[u]);
                nw[u] = n_window(S.nmask, g[u]);
                pm[u] = flags[u] ? O.group_fields[group[u] * kGroupFields + (__ffs((int) flags[u]) - 1)] : -1;
            }
            int64_t r[kOrdU], beg[kOrdU], end[kOrdU];
            int W[kOrdU];
            int64_t toff[kOrdU];
#pragma unroll
            for (int u = 0; u < kOrdU; u++) {
                const int64_t o0 = S.offsets[lo_r[u]], o1 = S.offsets[lo_r[u] + 1];
                const int64_t o2 = lo_r[u] + 2 <= S.R ? S.offsets[lo_r[u] + 2] : o1;
                if (g[u] < o1) { r[u] = lo_r[u]; beg[u] = o0; end[u] = o1; }
                else if (g[u] < o2) { r[u] = lo_r[u] + 1; beg[u] = o1; end[u] = o2; }
                else { r[u] = find_region(S, g[u]); beg[u] = S.offsets[r[u]]; end[u] = S.offsets[r[u] + 1]; }
                W[u] = pm[u] >= 0 ? Pw.width[pm[u]] : 0;
                toff[u] = pm[u] >= 0 ? Pw.tab_off[pm[u]] : 0;
            }
#pragma unroll
            for (int u = 0; u < kOrdU; u++) {
                if (!live[u]) continue;
                const int64_t gk = (int64_t) (((uint64_t) r[u] << O.pbits) | (uint64_t) (g[u] - beg[u]));
                bool first = true;
                while (flags[u]) {
                    const int field = __ffs((int) flags[u]) - 1;
                    flags[u] &= flags[u] - 1u;
                    int32_t m = pm[u];
                    int w = W[u];
                    int64_t to = toff[u];
                    if (!first) {
                        m = O.group_fields[group[u] * kGroupFields + field];
                        if (m >= 0) { w = Pw.width[m]; to = Pw.tab_off[m]; }
                    }
                    first = false;
                    if (m < 0) continue;
                    if (g[u] + w > end[u]) continue;                         // window runs past its region (cscore.c:340)
                    double fwd, rev;
                    if (w <= 32) score_window32(Pw.tab2 + to, w, cw[u], nw[u], fwd, rev);
                    else score_window(S, Pw.tab2 + to, w, g[u], fwd, rev);
                    // the reference's normalisation and threshold test, verbatim (cscore.c:356-358 / 373-375); raw_floor: see test_and_emit
                    const double floor_ = Pw.raw_floor[m];
                    const bool try_f = (O.strand_mask & 1) && !(fwd < floor_), try_r = (O.strand_mask & 2) && !(rev < floor_);
                    if (!try_f && !try_r) continue;
                    const double max_raw = Pw.max_raw[m], cutoff = Pw.cutoff[m];
#pragma unroll
                    for (int sd = 0; sd < 2; sd++) {
                        if (!(sd ? try_r : try_f)) continue;
                        const double sc = (sd ? rev : fwd) / max_raw;
                        if (sc - cutoff >= -1e-10) {
                            const uint32_t slot = atomicAdd(&s_n, 1u);
                            if (slot < (uint32_t) kOrdStage) {
                                st_key[slot] = ((uint64_t) (uint32_t) m << (O.gbits + 1)) | ((uint64_t) gk << 1) | (uint64_t) sd;
                                st_val[slot] = sc;
                                atomicAdd(&hist[m], 1u);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        uint32_t n = s_n;
        if (n > (uint32_t) kOrdStage) {                                // the chunk holds too many hits: the host halves the chunks
            if (tid == 0) { atomicOr(O.overflow, 2u); atomicMax(O.overflow + 2, n); }
            n = kOrdStage;
        }
        // ---- group by motif: row of the count matrix, group starts
        uint16_t *row = O.cnt_cm + c * O.P;
        for (int m = tid; m < O.P; m += 256) { const uint32_t h = hist[m]; row[m] = (uint16_t) h; start[m] = h; }
        __syncthreads();
        (void) block_excl_scan(start, O.P, s_scratch);
        if (tid == 0) {
            s_base = n ? atomicAdd(O.n_hits, (unsigned long long) n) : 0ULL;
            O.chunk_off[c] = s_base;
            O.chunk_n[c] = n;
        }
        for (uint32_t i = tid; i < n; i += 256) {
            const uint32_t m = (uint32_t) (st_key[i] >> (O.gbits + 1));
            order[start[m] + atomicAdd(&cursor[m], 1u)] = (uint16_t) i;
        }
        __syncthreads();
        // ---- rank inside the (chunk, motif) group by key, write the chunk's hits sorted
        const unsigned long long cbase = s_base;
        for (uint32_t p = tid; p < n; p += 256) {
            const uint32_t i = order[p];
            const uint64_t key = st_key[i];
            const uint32_t m = (uint32_t) (key >> (O.gbits + 1));
            const uint32_t a = start[m], cnt = hist[m];
            uint32_t rank = 0;
            for (uint32_t q = a; q < a + cnt; q++) rank += st_key[order[q]] < key ? 1u : 0u;
            if (cbase + a + rank < O.hit_cap) {
                O.keys[cbase + a + rank] = key;
                O.vals[cbase + a + rank] = st_val[i];
            }
        }
        __syncthreads();
    }
}

// ---- column-wise prefix over the count matrix cnt[n_chunks][P] (u16): segments of kOrdSeg chunks

// seg_sum[seg][m] = sum of cnt[c][m] over the segment's chunks.  grid = (ceil(P / 64), n_seg), 256 threads: 4 row lanes x 64 columns
__global__ void __launch_bounds__(256) ord_seg_sum_kernel(const uint16_t *__restrict__ cnt, int64_t n_chunks, int32_t P, uint32_t *__restrict__ seg_sum) {
    __shared__ uint32_t part[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int64_t c0 = (int64_t) blockIdx.y * kOrdSeg, c1 = c0 + kOrdSeg < n_chunks ? c0 + kOrdSeg : n_chunks;
    uint32_t s = 0;
    if (col < P) for (int64_t c = c0 + rl; c < c1; c += 4) s += cnt[c * P + col];
    part[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && col < P) seg_sum[(int64_t) blockIdx.y * P + col] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

// one WAVE per motif: seg_sum[.][m] -> exclusive prefix over the segments (in place; a motif has < 2^32 hits per call), motif_tot[m]
__global__ void __launch_bounds__(256) ord_seg_scan_kernel(uint32_t *__restrict__ seg_sum, int32_t n_seg, int32_t P, unsigned long long *__restrict__ motif_tot) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= P) return;
    const int per = (n_seg + 63) / 64, s0 = lane * per, s1 = s0 + per < n_seg ? s0 + per : n_seg;
    uint32_t local = 0;
    for (int s = s0; s < s1; s++) local += seg_sum[(int64_t) s * P + m];
    uint32_t tot;
    uint32_t run = wave_excl_scan(local, tot);
    for (int s = s0; s < s1; s++) { const uint32_t v = seg_sum[(int64_t) s * P + m]; seg_sum[(int64_t) s * P + m] = run; run += v; }
    if (lane == 0) motif_tot[m] = tot;
}

// motif_first = exclusive prefix of the motif totals ([P + 1]: the last entry is the number of hits).  ONE block of 256 threads.
__global__ void __launch_bounds__(256) ord_motif_first_kernel(const unsigned long long *__restrict__ motif_tot, int32_t P, int64_t *__restrict__ motif_first) {
    __shared__ unsigned long long wsum[4];
    const int tid = threadIdx.x, per = (P + 255) / 256;
    unsigned long long local = 0;
    for (int i = tid * per; i < P && i < (tid + 1) * per; i++) local += motif_tot[i];
    unsigned long long x = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long y = __shfl_up(x, o);
        if ((tid & 63) >= o) x += y;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = x;
    __syncthreads();
    unsigned long long run = x - local;
    for (int w = 0; w < (tid >> 6); w++) run += wsum[w];
    for (int i = tid * per; i < P && i < (tid + 1) * per; i++) { motif_first[i] = (int64_t) run; run += motif_tot[i]; }
    if (tid == 255) motif_first[P] = (int64_t) (wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

// off[c][m] = seg prefix + prefix inside the segment.  grid = (ceil(P / 64), n_seg), 64 threads: one column each, the segment's chunks in turn
__global__ void __launch_bounds__(64) ord_seg_prefix_kernel(const uint16_t *__restrict__ cnt, int64_t n_chunks, int32_t P, const uint32_t *__restrict__ seg_off,
                                                            uint32_t *__restrict__ off) {
    const int col = blockIdx.x * 64 + threadIdx.x;
    if (col >= P) return;
    const int64_t c0 = (int64_t) blockIdx.y * kOrdSeg, c1 = c0 + kOrdSeg < n_chunks ? c0 + kOrdSeg : n_chunks;
    uint32_t run = seg_off[(int64_t) blockIdx.y * P + col];
    int64_t c = c0;
    for (; c + 8 <= c1; c += 8) {                                      // 8 independent loads in flight
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = cnt[(c + j) * P + col];
#pragma unroll
        for (int j = 0; j < 8; j++) { off[(c + j) * P + col] = run; run += v[j]; }
    }
    for (; c < c1; c++) { off[c * P + col] = run; run += cnt[c * P + col]; }
}

// One block per chunk (looping): every hit of the chunk to its final rank; the key unpacked as finalize_rp_kernel does.
// Dynamic LDS: a [P] u32 (group starts inside the chunk) | dst [P] u64 (motif_first + column prefix)
__global__ void __launch_bounds__(256) ord_place_kernel(const OrdArgs O, const uint32_t *__restrict__ off, const int64_t *__restrict__ motif_first,
                                                        int64_t *__restrict__ seq_idx, int64_t *__restrict__ pos, double *__restrict__ score,
                                                        int8_t *__restrict__ strand, int rbits, uint64_t out_cap) {
    extern __shared__ unsigned char ord_lds[];
    uint32_t *a = reinterpret_cast<uint32_t *>(ord_lds);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(a + ((O.P + 1) & ~1));
    __shared__ uint32_t s_scratch[8];
    const int tid = threadIdx.x;
    const uint64_t rmask = (1ULL << rbits) - 1ULL, pmask = (1ULL << O.pbits) - 1ULL, gmask = (1ULL << O.gbits) - 1ULL;
    for (int64_t c = blockIdx.x; c < O.n_chunks; c += gridDim.x) {
        const uint32_t n = O.chunk_n[c];
        if (n == 0) continue;                                          // (block-uniform)
        const unsigned long long cbase = O.chunk_off[c];
        for (int m = tid; m < O.P; m += 256) {
            a[m] = O.cnt_cm[c * O.P + m];
            dst[m] = (unsigned long long) motif_first[m] + off[c * O.P + m];
        }
        __syncthreads();
        (void) block_excl_scan(a, O.P, s_scratch);
        for (uint32_t i = tid; i < n; i += 256) {
            if (cbase + i >= O.hit_cap) break;                         // (the hit list overflowed: the host runs the pass again)
            const uint64_t key = O.keys[cbase + i];
            const uint32_t m = (uint32_t) (key >> (O.gbits + 1));
            const unsigned long long d = dst[m] + (i - a[m]);
            if (d >= out_cap) continue;                                // (more hits than predicted: the host runs the pass again)
            const uint64_t coord = (key >> 1) & gmask;
            seq_idx[d] = (int64_t) ((coord >> O.pbits) & rmask);
            pos[d] = (int64_t) (coord & pmask);
            score[d] = O.vals[cbase + i];
            strand[d] = (int8_t) ((key & 1ULL) ? 2 : 1);
        }
        __syncthreads();
    }
}

// regions with >= 1 hit per motif (stats.py:29-31), from the placed arrays: one atomic per (wave, motif)
__global__ void __launch_bounds__(256) ord_region_counts_kernel(const int64_t *__restrict__ motif_first, int32_t P, uint64_t out_cap,
                                                                const int64_t *__restrict__ seq_idx, unsigned long long *__restrict__ region_counts) {
    int64_t n = motif_first[P];
    if ((uint64_t) n > out_cap) n = (int64_t) out_cap;
    for (int64_t i0 = (int64_t) blockIdx.x * blockDim.x; i0 < n; i0 += (int64_t) gridDim.x * blockDim.x) {
        const int64_t i = i0 + threadIdx.x;
        const bool live = i < n;
        int32_t m = -1;
        bool new_pair = false;
        if (live) {
            int32_t lo = 0, hi = P;                                    // motif_first[lo] <= i < motif_first[hi]
            while (hi - lo > 1) { const int32_t mid = (lo + hi) >> 1; if (motif_first[mid] <= i) lo = mid; else hi = mid; }
            m = lo;
            new_pair = i == motif_first[m] || seq_idx[i] != seq_idx[i - 1];
        }
        unsigned long long todo = __ballot(live && new_pair);
        while (todo) {
            const int leader = __ffsll((long long) todo) - 1;
            const int32_t mm = __shfl(m, leader);
            const unsigned long long same = __ballot(live && new_pair && m == mm);
            if ((int) (threadIdx.x & 63) == leader) atomicAdd(&region_counts[mm], (unsigned long long) __popcll(same));
            todo &= ~same;
        }
    }
}

size_t ord_rescore_lds_bytes(int32_t P, int n_src) {
    return (size_t) kOrdStage * 16 + (size_t) P * 12 + ((size_t) n_src + 2) * 4 + (size_t) kOrdStage * 2 + 64;
}

int launch_rescore_ordered(const OrdArgs &O, int n_blocks, hipStream_t st) {
    const size_t lds = ord_rescore_lds_bytes(O.P, O.n_tiles * O.k);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        MS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(rescore_ordered_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        lds_set = lds;
    }
    const int64_t nb = std::max<int64_t>(1, std::min<int64_t>(n_blocks, O.n_chunks));
    hipLaunchKernelGGL(rescore_ordered_kernel, dim3((unsigned) nb), dim3(256), lds, st, O);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

// count matrix -> per-(chunk, motif) offsets and per-motif offsets; then every hit to its place, then the region counts.
// seg_sum: [n_seg][P] u32 work space; off: [n_chunks][P] u32 work space; motif_tot: [P] u64 work space.
int launch_ordered_place(const OrdArgs &O, uint32_t *seg_sum, uint32_t *off, unsigned long long *motif_tot, int64_t *motif_first,
                         int64_t *seq_idx, int64_t *pos, double *score, int8_t *strand, uint64_t out_cap, int rbits, int n_cu, hipStream_t st) {
    const int n_seg = (int) ((O.n_chunks + kOrdSeg - 1) / kOrdSeg);
    const unsigned ct = (unsigned) ((O.P + 63) / 64);
    hipLaunchKernelGGL(ord_seg_sum_kernel, dim3(ct, (unsigned) n_seg), dim3(256), 0, st, O.cnt_cm, O.n_chunks, O.P, seg_sum);
    MS_HIP(hipGetLastError());
    hipLaunchKernelGGL(ord_seg_scan_kernel, dim3((unsigned) ((O.P + 3) / 4)), dim3(256), 0, st, seg_sum, n_seg, O.P, motif_tot);
    MS_HIP(hipGetLastError());
    hipLaunchKernelGGL(ord_motif_first_kernel, dim3(1), dim3(256), 0, st, motif_tot, O.P, motif_first);
    MS_HIP(hipGetLastError());
    hipLaunchKernelGGL(ord_seg_prefix_kernel, dim3(ct, (unsigned) n_seg), dim3(64), 0, st, O.cnt_cm, O.n_chunks, O.P, seg_sum, off);
    MS_HIP(hipGetLastError());
    const size_t lds = (size_t) ((O.P + 1) & ~1) * 4 + (size_t) O.P * 8 + 64;
    const int64_t nb = std::max<int64_t>(1, std::min<int64_t>((int64_t) n_cu * 8, O.n_chunks));
    hipLaunchKernelGGL(ord_place_kernel, dim3((unsigned) nb), dim3(256), lds, st, O, off, motif_first, seq_idx, pos, score, strand, rbits, out_cap);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

int launch_ordered_region_counts(const int64_t *motif_first, int32_t P, uint64_t out_cap, const int64_t *seq_idx, unsigned long long *region_counts,
                                 int n_cu, hipStream_t st) {
    hipLaunchKernelGGL(ord_region_counts_kernel, dim3((unsigned) (n_cu * 8)), dim3(256), 0, st, motif_first, P, out_cap, seq_idx, region_counts);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

}  // namespace ms
