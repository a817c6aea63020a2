// ms_regions.hip -- scan-once for region lists that OVERLAP (ms_scan_regions_once).
//
// The reference scans every region on its own (scanner.py:71-87 cuts them, cscore.c:336-390 scans each), so a base that
// lies in k regions is scored k times: peaks +- window/2 closer together than the window, and the 5x random control set
// drawn around them (cli/scan.py:43-48, 76-86; region/utils.py:89-145), overlap heavily.  Here the union of the regions is
// scanned ONCE: overlapping regions of a chromosome are merged into spans on the host, the spans are cut from the resident
// genome and scanned as a sequence set, and every span hit is handed to each region that contains all of its bases
// (cscore.c:340: a window must lie inside its sequence), re-keyed as (motif, region, position in the region, strand) and put
// into the reference's order by the same sort + finalize the plain scan uses.  Scores are the span scan's fp64 scores: the
// window's bases are the same bases, so they are bit-identical to what a per-region scan computes.
#include <algorithm>
#include <numeric>

#include "ms_handles.h"

namespace ms {

namespace {

struct RegionIndex {                 // device view of the merged layout
    const int64_t *span_first;       // [S+1] first sorted region of every span
    const int64_t *rel_start;        // [R] region start relative to its span's begin, ascending inside a span
    const int64_t *rel_end;          // [R]
    const int64_t *orig;             // [R] index of the region in the caller's list
    const int64_t *span_maxlen;      // [S] longest region of the span
};

// sorted regions [lo, hi] of span s that can hold a site at [g, g + W): start <= g, and start > g + W - maxlen
__device__ __forceinline__ void candidate_range(const RegionIndex &X, int64_t s, int64_t g, int W, int64_t &lo, int64_t &hi) {
    const int64_t jb = X.span_first[s], je = X.span_first[s + 1];
    int64_t a = jb, b = je;                                   // first j with rel_start[j] > g
    while (a < b) { const int64_t m = (a + b) >> 1; if (X.rel_start[m] <= g) a = m + 1; else b = m; }
    hi = a - 1;
    const int64_t need = g + W - X.span_maxlen[s];            // a region starting before this cannot reach g + W
    a = jb; b = je;                                           // first j with rel_start[j] >= need
    while (a < b) { const int64_t m = (a + b) >> 1; if (X.rel_start[m] < need) a = m + 1; else b = m; }
    lo = a;
}

// an unordered span hit: key = motif << (gbits + 1) | coordinate << 1 | strand bit (ms_handles.h, MS_SCAN_RAW_INTERNAL)
struct RawKeys {
    const uint64_t *keys;
    const double *vals;
    int gbits, pbits;
    const int64_t *span_off;         // [S+1] span offsets in the scanned sequence set (pbits == 0: global positions)
    int64_t S;
};

__device__ __forceinline__ void raw_decode(const RawKeys &K, uint64_t k, int32_t &m, int64_t &s, int64_t &g, uint64_t &sbit) {
    sbit = k & 1ULL;
    m = (int32_t) (k >> (K.gbits + 1));
    const uint64_t coord = (k >> 1) & ((1ULL << K.gbits) - 1ULL);
    if (K.pbits) {
        s = (int64_t) (coord >> K.pbits);
        g = (int64_t) (coord & ((1ULL << K.pbits) - 1ULL));
    } else {
        int64_t lo = 0, hi = K.S;                             // span_off[lo] <= coord < span_off[hi]
        while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if ((uint64_t) K.span_off[mid] <= coord) lo = mid; else hi = mid; }
        s = lo;
        g = (int64_t) coord - K.span_off[lo];
    }
}

__global__ void __launch_bounds__(256) once_count_kernel(int64_t n, const RawKeys K, const int32_t *__restrict__ width, const RegionIndex X,
                                                         uint32_t *__restrict__ cnt) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t m; int64_t s, g; uint64_t sbit;
    raw_decode(K, K.keys[i], m, s, g, sbit);
    const int W = width[m];
    int64_t lo, hi;
    candidate_range(X, s, g, W, lo, hi);
    uint32_t c = 0;
    for (int64_t j = lo; j <= hi; j++) c += X.rel_end[j] >= g + W ? 1u : 0u;
    cnt[i] = c;
}

__global__ void __launch_bounds__(256) once_expand_kernel(int64_t n, const RawKeys K, const int32_t *__restrict__ width, const RegionIndex X,
                                                          const uint64_t *__restrict__ dst, int rbits, int pbits,
                                                          uint64_t *__restrict__ keys, double *__restrict__ vals) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t m; int64_t s, g; uint64_t sbit;
    raw_decode(K, K.keys[i], m, s, g, sbit);
    const int W = width[m];
    int64_t lo, hi;
    candidate_range(X, s, g, W, lo, hi);
    uint64_t d = dst[i];
    const double sc = K.vals[i];
    for (int64_t j = lo; j <= hi; j++)
        if (X.rel_end[j] >= g + W) {
            keys[d] = ((uint64_t) m << (rbits + pbits + 1)) | ((uint64_t) X.orig[j] << (pbits + 1)) | ((uint64_t) (g - X.rel_start[j]) << 1) | sbit;
            vals[d] = sc;
            d++;
        }
}

}  // namespace

// ---- counts only, without ordering the hits (round 6) ----
// What the enrichment statistics read of a region set -- per motif the number of regions with >= 1 site (stats.py:29-31) and the number of
// sites -- from the UNORDERED hit keys the fp64 stage leaves (key = motif << (gbits + 1) | (region << pbits | position) << 1 | strand bit):
// one pass sets a flag BYTE per (motif, region) with a plain store (a bit map needs an atomic per hit, and scattered global atomics run at
// ~22 G/s chip-wide, returning or not: 0.7 ms for the 15 M hits of a 250 000-region batch -- as long as the radix sort they were meant to
// save; tools/counts_only_time.py), per-motif site numbers aggregated per wave and block in LDS; a second pass counts every motif's flags.
// No radix sort, no finalize, no hit arrays.
constexpr int kCountBins = 4096;                              // motifs the LDS histogram holds (more: the ordered path)

__global__ void __launch_bounds__(256) count_only_kernel(const uint64_t *__restrict__ keys, int64_t n, const unsigned long long *__restrict__ n_dev, int gbits, int pbits,
                                                         int64_t row_words, int32_t P, uint32_t *__restrict__ bitmap, unsigned long long *__restrict__ motif_hits) {
    __shared__ unsigned int h_hits[kCountBins];
    for (int i = threadIdx.x; i < P; i += blockDim.x) h_hits[i] = 0;
    __syncthreads();
    if (n_dev) { const unsigned long long nd = *n_dev; if ((unsigned long long) n > nd) n = (int64_t) nd; }
    const uint64_t cmask = (1ULL << gbits) - 1ULL;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    const int64_t n_round = ((n + stride - 1) / stride) * stride;          // whole waves take every trip: the aggregation below is wave-wide
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        const uint64_t key = i < n ? keys[i] : ~0ULL;
        const uint32_t motif = (uint32_t) (key >> (gbits + 1));
        const bool live = i < n && motif < (uint32_t) P;     // (an all-ones padding key of a predicted-size list: never below n, but harmless)
        if (live) {
            const uint64_t region = ((key >> 1) & cmask) >> pbits;
            // a plain one-byte store, no atomic (every writer stores the same 1; the L2s merge partial lines by byte mask)
            reinterpret_cast<volatile uint8_t *>(bitmap)[(uint64_t) motif * (uint64_t) row_words * 4ULL + region] = (uint8_t) 1;
        }
        // the fp64 stage emits its hits in motif-ordered chunks: a wave's 64 keys hold one or two motifs, and 64 LDS atomics on one address
        // serialise.  So the wave counts per DISTINCT motif: one leader lane per motif adds the popcount.
        unsigned long long todo = __builtin_amdgcn_ballot_w64(live);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const uint32_t m0 = (uint32_t) __builtin_amdgcn_readlane((int) motif, leader);
            const unsigned long long same = __builtin_amdgcn_ballot_w64(live && motif == m0);
            if ((int) (threadIdx.x & 63u) == leader) atomicAdd(&h_hits[m0], (unsigned int) __popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += blockDim.x)
        if (h_hits[i]) atomicAdd(&motif_hits[i], (unsigned long long) h_hits[i]);
}

// one block per motif: the regions with a site = the set bits of the motif's row
__global__ void __launch_bounds__(256) count_rows_kernel(const uint32_t *__restrict__ bitmap, int64_t row_words, unsigned long long *__restrict__ region_counts) {
    __shared__ unsigned long long part[256];
    const uint32_t *row = bitmap + (uint64_t) blockIdx.x * (uint64_t) row_words;
    unsigned long long c = 0;
    for (int64_t w = threadIdx.x; w < row_words; w += blockDim.x) c += (unsigned long long) __popc(row[w] & 0x01010101u);      // four flag bytes per word
    part[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int) threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) region_counts[blockIdx.x] = part[0];
}

// motif_first[0 .. P] = exclusive prefix of the per-motif site numbers (one block; P <= kCountBins)
__global__ void __launch_bounds__(256) motif_prefix_kernel(const unsigned long long *__restrict__ motif_hits, int32_t P, int64_t *__restrict__ motif_first) {
    __shared__ unsigned long long part[256];
    const int per = (P + 255) / 256, lo = threadIdx.x * per, hi = lo + per < P ? lo + per : P;
    unsigned long long sum = 0;
    for (int i = lo; i < hi; i++) sum += motif_hits[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    unsigned long long run = 0;
    for (int t = 0; t < (int) threadIdx.x; t++) run += part[t];
    for (int i = lo; i < hi; i++) { motif_first[i] = (int64_t) run; run += motif_hits[i]; }
    if (threadIdx.x == 255) motif_first[P] = (int64_t) run;     // (the last thread's range ends at P, or is empty and `run` is the total)
}

bool count_only_supported(int32_t P, int64_t R, int pbits) { return pbits > 0 && P > 0 && P <= kCountBins && (double) P * (double) (R + 16) <= 6.0e9; }
size_t count_only_bitmap_words(int32_t P, int64_t R) { return (size_t) P * (size_t) ((R + 3) / 4); }      // (a byte per (motif, region), rows of whole words)

int launch_count_only(const uint64_t *keys, int64_t n, const unsigned long long *n_dev, int gbits, int pbits, int64_t R, int32_t P, uint32_t *bitmap,
                      unsigned long long *region_counts, unsigned long long *motif_hits, int64_t *motif_first, hipStream_t st) {
    const int64_t row_words = (R + 3) / 4;
    MS_HIP(hipMemsetAsync(bitmap, 0, count_only_bitmap_words(P, R) * sizeof(uint32_t), st));
    MS_HIP(hipMemsetAsync(motif_hits, 0, (size_t) P * sizeof(unsigned long long), st));
    if (n > 0) {
        const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(2048, (n + 4095) / 4096));
        hipLaunchKernelGGL(count_only_kernel, dim3((unsigned) blocks), dim3(256), 0, st, keys, n, n_dev, gbits, pbits, row_words, P, bitmap, motif_hits);
        MS_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(count_rows_kernel, dim3((unsigned) P), dim3(256), 0, st, bitmap, row_words, region_counts);
    MS_HIP(hipGetLastError());
    hipLaunchKernelGGL(motif_prefix_kernel, dim3(1), dim3(256), 0, st, motif_hits, P, motif_first);
    MS_HIP(hipGetLastError());
    return MS_OK;
}

}  // namespace ms

using namespace ms;

extern "C" int ms_scan_regions_once(const ms_pwmset *pwms_c, const ms_genome *g, const int32_t *chrom, const int64_t *start,
                                    const int64_t *end, int64_t n_regions, int strand_mask, uint32_t flags, ms_result **out) {
    if (!out) { set_error("out is NULL"); return MS_ERR_INVALID; }
    *out = nullptr;
    if (!pwms_c || !g) { set_error("NULL handle"); return MS_ERR_INVALID; }
    if (strand_mask < 1 || strand_mask > 3) { set_error("invalid strand mask %d (1 '+', 2 '-', 3 both)", strand_mask); return MS_ERR_INVALID; }
    if (n_regions < 0 || (n_regions > 0 && (!chrom || !start || !end))) { set_error("bad region arrays"); return MS_ERR_INVALID; }
    if (n_regions >= (1LL << 31)) { set_error("too many regions"); return MS_ERR_INVALID; }
    if (flags & ~(uint32_t) MS_SCAN_EXACT_ONLY) { set_error("unknown scan flags 0x%x", flags); return MS_ERR_INVALID; }
    ms_pwmset *pwms = const_cast<ms_pwmset *>(pwms_c);
    const ms_seqset *G = reinterpret_cast<const ms_seqset *>(g);
    const int64_t n_chroms = G->R;
    const int64_t *goff = G->offsets.data();
    const size_t R = (size_t) n_regions;

    // ---- host: order the regions by (chromosome, start), merge overlapping ones into spans --------------------------
    std::vector<int64_t> order(R), rel_start(R), rel_end(R), span_first, span_maxlen, sp_start, sp_end;
    std::vector<int32_t> sp_chrom;
    int64_t max_len = 0;
    try {
        for (size_t r = 0; r < R; r++) {
            const int64_t ch = chrom[r];
            if (ch < 0 || ch >= n_chroms) { set_error("region %zu: chromosome index %d out of range", r, chrom[r]); return MS_ERR_INVALID; }
            if (start[r] < 0 || end[r] < start[r] || end[r] > goff[ch + 1] - goff[ch]) {
                set_error("region %zu: [%lld, %lld) is outside chromosome %d of length %lld", r, (long long) start[r], (long long) end[r],
                          chrom[r], (long long) (goff[ch + 1] - goff[ch]));
                return MS_ERR_INVALID;
            }
            max_len = std::max(max_len, end[r] - start[r]);
        }
        // order by (chromosome, start, index): LSD radix sort of the packed key, 16 bits a pass, only the passes the data needs
        // (an indirect std::sort of 200k regions costs more than scanning them)
        {
            int64_t max_start = 0;
            for (size_t r = 0; r < R; r++) max_start = std::max(max_start, start[r]);
            int sbits = 1, cbits = 1;
            while ((1LL << sbits) <= max_start) sbits++;
            while ((1LL << cbits) < std::max<int64_t>(n_chroms, 1)) cbits++;
            std::vector<uint64_t> key(R), key2(R);
            std::vector<int64_t> order2(R);
            for (size_t r = 0; r < R; r++) { key[r] = ((uint64_t) chrom[r] << sbits) | (uint64_t) start[r]; order[r] = (int64_t) r; }
            for (int shift = 0; shift < sbits + cbits; shift += 16) {
                std::vector<size_t> count(65537, 0);
                for (size_t r = 0; r < R; r++) count[((key[r] >> shift) & 0xFFFFu) + 1]++;
                for (size_t b = 0; b < 65536; b++) count[b + 1] += count[b];
                for (size_t r = 0; r < R; r++) {
                    const size_t d = count[(key[r] >> shift) & 0xFFFFu]++;
                    key2[d] = key[r];
                    order2[d] = order[r];
                }
                key.swap(key2);
                order.swap(order2);
            }
        }
        for (size_t j = 0; j < R; j++) {
            const int64_t r = order[j];
            const bool joins = !sp_chrom.empty() && sp_chrom.back() == chrom[r] && start[r] < sp_end.back();     // true overlap only
            if (!joins) {
                sp_chrom.push_back(chrom[r]);
                sp_start.push_back(start[r]);
                sp_end.push_back(end[r]);
                span_first.push_back((int64_t) j);
                span_maxlen.push_back(end[r] - start[r]);
            } else {
                sp_end.back() = std::max(sp_end.back(), end[r]);
                span_maxlen.back() = std::max(span_maxlen.back(), end[r] - start[r]);
            }
            rel_start[j] = start[r] - sp_start.back();
            rel_end[j] = end[r] - sp_start.back();
        }
        span_first.push_back((int64_t) R);
    } catch (const std::bad_alloc &) { set_error("out of host memory"); return MS_ERR_NOMEM; }
    const size_t S = sp_chrom.size();

    // ---- device: cut the spans, scan them as a sequence set -------------------------------------------------------------
    ms_seqset *spans = nullptr;
    int rc = ms_seqset_from_genome(g, sp_chrom.data(), sp_start.data(), sp_end.data(), (int64_t) S, &spans);
    if (rc) return rc;
    DeviceCtx *c;
    if ((rc = get_ctx(spans->device, &c))) { ms_seqset_free(spans); return rc; }
    std::lock_guard<std::mutex> lk_dev(c->mu);
    std::lock_guard<std::mutex> lk_pwm(pwms->mu);
    // the span hits are re-keyed anyway: take them unordered from the scan's scratch (no sort / unpack of the span result)
    ms_result *r1 = nullptr;
    rc = scan_locked(c, pwms, spans, strand_mask, flags | MS_SCAN_RAW_INTERNAL, &r1);
    const int64_t union_bases = spans->n_bases;
    if (rc) { ms_seqset_free(spans); return rc; }
    auto fail = [&](int code) { ms_result_free(r1); ms_seqset_free(spans); return code; };
    if ((rc = pwmset_upload(pwms, c->device, c->stream))) return fail(rc);
    const size_t n1 = (size_t) r1->n_hits;
    RawKeys K;
    K.keys = c->sc.keys; K.vals = c->sc.vals; K.gbits = r1->raw_gbits; K.pbits = r1->raw_pbits; K.span_off = spans->d_offsets; K.S = spans->R;

    int rbits = 1, pbits = 1, mbits = 1;
    while ((1LL << rbits) < std::max<int64_t>(n_regions, 1)) rbits++;
    while ((1LL << pbits) < std::max<int64_t>(max_len, 1)) pbits++;
    while ((1 << mbits) < std::max(pwms->P, 1)) mbits++;
    if (mbits + rbits + pbits + 1 > 64) { set_error("regions too many / too long for one call"); return fail(MS_ERR_INVALID); }

    std::unique_ptr<ms_result> res(new (std::nothrow) ms_result());
    if (!res) { set_error("out of host memory"); return fail(MS_ERR_NOMEM); }
    res->device = r1->device;
    res->P = pwms->P;
    res->R = n_regions;
    res->stats = r1->stats;
    res->invalid = r1->invalid;
    res->motif_offsets.assign((size_t) pwms->P + 1, 0);
    ms_result *raw = res.release();
    auto fail2 = [&](int code) { ms_result_free(raw); return fail(code); };

    // index arrays + work buffers in one pooled block
    void *iblk = nullptr, *wblk = nullptr, *d_tmp = nullptr;
    size_t igot = 0, wgot = 0, tgot = 0;
    auto cleanup = [&]() { if (iblk) pool_free(c, iblk, igot); if (wblk) pool_free(c, wblk, wgot); if (d_tmp) pool_free(c, d_tmp, tgot); };
    auto up8 = [](size_t x) { return (x + 31) & ~(size_t) 31; };
    const size_t n_idx = up8(S + 1) + 3 * up8(R) + up8(S);
    if ((rc = pool_alloc(c, 8 * n_idx + 12 * up8(n1) + 256, &iblk, &igot))) { cleanup(); return fail2(rc); }
    int64_t *d_span_first = static_cast<int64_t *>(iblk);
    int64_t *d_rel_start = d_span_first + up8(S + 1), *d_rel_end = d_rel_start + up8(R), *d_orig = d_rel_end + up8(R);
    int64_t *d_maxlen = d_orig + up8(R);
    uint64_t *d_dst = reinterpret_cast<uint64_t *>(d_maxlen + up8(S));
    uint32_t *d_cnt = reinterpret_cast<uint32_t *>(d_dst + up8(n1));
    hipError_t he = hipMemcpyAsync(d_span_first, span_first.data(), 8 * (S + 1), hipMemcpyHostToDevice, c->stream);
    if (he == hipSuccess && R) he = hipMemcpyAsync(d_rel_start, rel_start.data(), 8 * R, hipMemcpyHostToDevice, c->stream);
    if (he == hipSuccess && R) he = hipMemcpyAsync(d_rel_end, rel_end.data(), 8 * R, hipMemcpyHostToDevice, c->stream);
    if (he == hipSuccess && R) he = hipMemcpyAsync(d_orig, order.data(), 8 * R, hipMemcpyHostToDevice, c->stream);
    if (he == hipSuccess && S) he = hipMemcpyAsync(d_maxlen, span_maxlen.data(), 8 * S, hipMemcpyHostToDevice, c->stream);
    if (he != hipSuccess) { cleanup(); set_error("index upload failed: %s", hipGetErrorString(he)); return fail2(MS_ERR_RUNTIME); }
    RegionIndex X;
    X.span_first = d_span_first; X.rel_start = d_rel_start; X.rel_end = d_rel_end; X.orig = d_orig; X.span_maxlen = d_maxlen;

    (void) hipEventRecord(c->ev[0], c->stream);
    uint64_t total = 0;
    if (n1 > 0) {
        hipLaunchKernelGGL(once_count_kernel, dim3((unsigned) ((n1 + 255) / 256)), dim3(256), 0, c->stream, (int64_t) n1, K, pwms->d_width, X, d_cnt);
        size_t tmp_bytes = 0;
        rc = exclusive_sum_u32(nullptr, &tmp_bytes, d_cnt, d_dst, n1, c->stream);
        if (!rc) rc = pool_alloc(c, tmp_bytes ? tmp_bytes : 256, &d_tmp, &tgot);
        if (!rc) rc = exclusive_sum_u32(d_tmp, &tmp_bytes, d_cnt, d_dst, n1, c->stream);
        uint32_t last_cnt = 0;
        uint64_t last_dst = 0;
        if (!rc) {
            he = hipMemcpyAsync(&last_cnt, d_cnt + (n1 - 1), 4, hipMemcpyDeviceToHost, c->stream);
            if (he == hipSuccess) he = hipMemcpyAsync(&last_dst, d_dst + (n1 - 1), 8, hipMemcpyDeviceToHost, c->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
            if (he != hipSuccess) { set_error("hand-out count failed: %s", hipGetErrorString(he)); rc = MS_ERR_RUNTIME; }
        }
        if (rc) { cleanup(); return fail2(rc); }
        total = last_dst + last_cnt;
    }
    raw->n_hits = (int64_t) total;
    {
        void *blk = nullptr;
        size_t got = 0;
        if ((rc = pool_alloc(c, result_block_bytes(pwms->P, (size_t) total), &blk, &got))) { cleanup(); return fail2(rc); }
        raw->block = blk;
        raw->block_bytes = got;
        result_carve(raw, blk, (size_t) total);
        he = hipMemsetAsync(raw->d_region_counts, 0, 8 * ((size_t) pwms->P + 1), c->stream);
        if (he != hipSuccess) { cleanup(); set_error("memset failed: %s", hipGetErrorString(he)); return fail2(MS_ERR_RUNTIME); }
    }
    if (total > 0) {
        const size_t nt = up8((size_t) total);
        size_t sort_bytes = 0;
        if ((rc = sort_hit_pairs(nullptr, &sort_bytes, nullptr, nullptr, nullptr, nullptr, (size_t) total, 0, mbits + rbits + pbits + 1, c->stream))) { cleanup(); return fail2(rc); }
        if ((rc = pool_alloc(c, 24 * nt + sort_bytes + 256, &wblk, &wgot))) { cleanup(); return fail2(rc); }
        uint64_t *d_keys = static_cast<uint64_t *>(wblk), *d_keys_sorted = d_keys + nt;
        double *d_vals = reinterpret_cast<double *>(d_keys_sorted + nt);
        void *d_sort_tmp = d_vals + nt;
        hipLaunchKernelGGL(once_expand_kernel, dim3((unsigned) ((n1 + 255) / 256)), dim3(256), 0, c->stream, (int64_t) n1, K, pwms->d_width, X,
                           d_dst, rbits, pbits, d_keys, d_vals);
        he = hipGetLastError();
        if (he != hipSuccess) { cleanup(); set_error("hand-out kernel failed: %s", hipGetErrorString(he)); return fail2(MS_ERR_RUNTIME); }
        if ((rc = sort_hit_pairs(d_sort_tmp, &sort_bytes, d_keys, d_keys_sorted, d_vals, raw->d_score, (size_t) total,
                                 0, mbits + rbits + pbits + 1, c->stream))) { cleanup(); return fail2(rc); }
        DevSeq none{};
        if ((rc = launch_finalize(d_keys_sorted, (int64_t) total, nullptr, rbits + pbits, rbits, pbits, pwms->P, none, raw->d_seq_idx, raw->d_pos,
                                  raw->d_strand, raw->d_motif_first, raw->d_region_counts, c->stream))) { cleanup(); return fail2(rc); }
    } else {
        DevSeq none{};
        if ((rc = launch_finalize(nullptr, 0, nullptr, rbits + pbits, rbits, pbits, pwms->P, none, raw->d_seq_idx, raw->d_pos, raw->d_strand,
                                  raw->d_motif_first, raw->d_region_counts, c->stream))) { cleanup(); return fail2(rc); }
    }
    (void) hipEventRecord(c->ev[1], c->stream);
    he = hipMemcpyAsync(raw->motif_offsets.data(), raw->d_motif_first, raw->motif_offsets.size() * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    cleanup();
    if (he != hipSuccess) { set_error("hand-out failed: %s", hipGetErrorString(he)); return fail2(MS_ERR_RUNTIME); }

    float ms01 = 0;
    (void) hipEventElapsedTime(&ms01, c->ev[0], c->ev[1]);
    ms_scan_stats &stt = raw->stats;
    stt.ms_finalize += ms01;                                     // the hand-out (count + prefix sum + expand + order + coordinates)
    stt.ms_total += ms01;
    stt.n_hits = (int64_t) total;
    stt.n_bases = union_bases;                                   // bases scanned: the union, each once (sum over regions: below)
    stt.n_windows = 0;                                           // the reference's unit count: every region scanned on its own
    {
        std::vector<int64_t> lens(R), suffix(R + 1, 0);
        for (size_t r = 0; r < R; r++) lens[r] = end[r] - start[r];
        std::sort(lens.begin(), lens.end());
        for (size_t r = R; r-- > 0;) suffix[r] = suffix[r + 1] + lens[r];
        for (int32_t p = 0; p < pwms->P; p++) {
            const int64_t W = pwms->widths[p];
            const size_t k = (size_t) (std::lower_bound(lens.begin(), lens.end(), W) - lens.begin());      // regions with L >= W
            stt.n_windows += suffix[k] - (int64_t) (R - k) * (W - 1);
        }
    }
    stt.hbm_bytes_algorithmic += 16 * ((int64_t) total - (int64_t) n1);
    ms_result_free(r1);
    ms_seqset_free(spans);
    *out = raw;
    return MS_OK;
}
