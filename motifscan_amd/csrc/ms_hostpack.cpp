// ms_hostpack.cpp -- convert_seq (cscore.c:81-114) and the region hints on HOST threads, for the batch stream's upload stage.
//
// Why (round 5): in a batch stream every KERNEL of the upload stage (pack_kernel, blk2reg_kernel) is launched beside the scan stage's
// pre-filter, whose persistent blocks fill every CU -- a side kernel gets a CU only when one of them retires, so the upload
// stage "works" 42-55 ms per configs[3] pass although its copies take 18 and its kernels 0.6, and with every hit copied out it is the
// stage the scan waits for.  Packed on the host the stage has NO kernel: 0.375 B/base of codes + mask and 0.31 B/base of hints cross the
// link instead of 1 B/base of ASCII, on the copy engines, whatever the CUs do.
//
// Layout = the device kernels' (ms_kernels.hip: pack_kernel, blk2reg_kernel), word for word: a unit of 32 bases -> codes[2u], codes[2u+1]
// (2 bits per base, base i at bits [2i, 2i+2): a/A 0, c/C 1, g/G 2, t/T 3, anything else 0) and nmask[u] (bit i: base i is none of those);
// positions past the end are code 0 / not N.  blk2reg[b] = the region of position 64 b (the last region r with offsets[r] <= 64 b);
// blkinfo[b] = {r, offsets[r] - 64 b, offsets[r+1] - 64 b, offsets[r+2] - 64 b} (clamped to offsets[R]), or {-1, 0, 0, 0} when a value
// does not fit 32 bits.
#include <immintrin.h>

#include <algorithm>
#include <cstdint>
#include <cstring>

namespace ms {

namespace {

inline void pack_unit_scalar(const uint8_t *p, int valid, uint32_t *c0, uint32_t *c1, uint32_t *nw) {
    uint64_t cw = 0;
    uint32_t n = 0;
    for (int i = 0; i < valid; i++) {
        const uint32_t ch = (uint32_t) p[i] | 0x20u;                          // fold case (cscore.c:93-108)
        const bool acgt = ch == 0x61u || ch == 0x63u || ch == 0x67u || ch == 0x74u;
        const uint32_t code = ((ch >> 1) ^ (ch >> 2)) & 3u;                   // a, c, g, t -> 0, 1, 2, 3
        cw |= (uint64_t) (acgt ? code : 0u) << (2 * i);
        n |= (acgt ? 0u : 1u) << i;
    }
    *c0 = (uint32_t) cw;
    *c1 = (uint32_t) (cw >> 32);
    *nw = n;
}

__attribute__((target("avx2,bmi2"))) void pack_units_avx2(const uint8_t *bases, int64_t u0, int64_t u1, uint32_t *codes, uint32_t *nmask) {
    const __m256i fold = _mm256_set1_epi8(0x20), a = _mm256_set1_epi8('a'), c = _mm256_set1_epi8('c'), g = _mm256_set1_epi8('g'), t = _mm256_set1_epi8('t');
    for (int64_t u = u0; u < u1; u++) {
        const __m256i v = _mm256_or_si256(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(bases + 32 * u)), fold);
        const __m256i is_c = _mm256_cmpeq_epi8(v, c), is_g = _mm256_cmpeq_epi8(v, g), is_t = _mm256_cmpeq_epi8(v, t);
        const uint32_t bit0 = (uint32_t) _mm256_movemask_epi8(_mm256_or_si256(is_c, is_t));       // c = 1, t = 3
        const uint32_t bit1 = (uint32_t) _mm256_movemask_epi8(_mm256_or_si256(is_g, is_t));       // g = 2, t = 3
        const uint32_t valid = (uint32_t) _mm256_movemask_epi8(_mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, a), is_c), _mm256_or_si256(is_g, is_t)));
        const uint64_t cw = _pdep_u64(bit0, 0x5555555555555555ULL) | _pdep_u64(bit1, 0xAAAAAAAAAAAAAAAAULL);
        codes[2 * u] = (uint32_t) cw;
        codes[2 * u + 1] = (uint32_t) (cw >> 32);
        nmask[u] = ~valid;
    }
}

}  // namespace

// units [u0, u1) of a sequence of n_bases bases (the caller splits the units over its threads)
void host_pack_units(const uint8_t *bases, int64_t n_bases, int64_t u0, int64_t u1, uint32_t *codes, uint32_t *nmask) {
    static const bool fast = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
    const int64_t full = std::min<int64_t>(u1, n_bases / 32);               // units that hold 32 real bases
    if (fast && full > u0) pack_units_avx2(bases, u0, full, codes, nmask);
    for (int64_t u = fast ? std::max(u0, full) : u0; u < u1; u++) {
        const int64_t left = n_bases - 32 * u;
        pack_unit_scalar(bases + 32 * u, (int) (left >= 32 ? 32 : (left > 0 ? left : 0)), &codes[2 * u], &codes[2 * u + 1], &nmask[u]);
    }
}

// blocks [b0, b1) of the region hints (n_blocks = (n_bases + 63) / 64 + 1 in all); info: four int32 per block
void host_region_hints(const int64_t *offsets, int64_t R, int64_t b0, int64_t b1, int32_t *blk2reg, int32_t *info, bool all_far) {
    if (b0 >= b1) return;
    // the region of position 64 b0: the last r in [0, R) with offsets[r] <= position (blk2reg_kernel's find_region_bsearch), then a merge
    int64_t lo = 0, hi = R;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= b0 * 64) lo = mid; else hi = mid;
    }
    int64_t r = lo;
    for (int64_t b = b0; b < b1; b++) {
        const int64_t base = b * 64;
        while (r + 1 < R && offsets[r + 1] <= base) r++;
        blk2reg[b] = (int32_t) r;
        const int64_t o0 = offsets[r] - base, o1 = offsets[r + 1 <= R ? r + 1 : R] - base, o2 = offsets[r + 2 <= R ? r + 2 : (r + 1 <= R ? r + 1 : R)] - base;
        const bool fits = o0 > -(1LL << 31) && o1 < (1LL << 31) && o2 < (1LL << 31) && o1 > -(1LL << 31) && o2 > -(1LL << 31) && r < (1LL << 31);
        int32_t *q = info + 4 * b;
        if (fits && !all_far) { q[0] = (int32_t) r; q[1] = (int32_t) o0; q[2] = (int32_t) o1; q[3] = (int32_t) o2; }
        else { q[0] = -1; q[1] = q[2] = q[3] = 0; }
    }
}

}  // namespace ms
