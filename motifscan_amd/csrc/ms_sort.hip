// ms_sort.hip -- ordering of the sparse hit list (keys = motif | position | strand bit,
// values = fp64 scores).  The hit list is ~1e-4 of the scanned units, so this is not the hot
// kernel; rocPRIM's device radix sort (AMD's own header-only primitives) is used as is.
// Kept in its own translation unit because the rocPRIM headers dominate compile time.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "ms_internal.h"

namespace ms {

int sort_hit_pairs(void *temp, size_t *temp_bytes, const uint64_t *keys_in, uint64_t *keys_out,
                   const double *vals_in, double *vals_out, size_t n, int begin_bit, int end_bit, hipStream_t stream) {
    if (end_bit < 1) end_bit = 1;
    if (end_bit > 64) end_bit = 64;
    if (begin_bit < 0 || begin_bit >= end_bit) begin_bit = 0;
    // (rocPRIM's gfx950 configuration for 8-byte keys with 8-byte values: 1024 threads x 8 items, 8 bits per pass, "match" ranking.
    // Ten bits per pass -- four passes instead of five over a scan's 38 ... 41 key bits -- measured slower: 3.25 against 2.81 ms for
    // 6.2e7 pairs, profiles/r03l_sort_bits.log.)
    MS_HIP(rocprim::radix_sort_pairs(temp, *temp_bytes, keys_in, keys_out, vals_in, vals_out, n, (unsigned int) begin_bit, (unsigned int) end_bit, stream));
    return MS_OK;
}

int sort_keys(void *temp, size_t *temp_bytes, const uint64_t *keys_in, uint64_t *keys_out, size_t n, int end_bit, hipStream_t stream) {
    if (end_bit < 1) end_bit = 1;
    if (end_bit > 64) end_bit = 64;
    MS_HIP(rocprim::radix_sort_keys(temp, *temp_bytes, keys_in, keys_out, n, 0u, (unsigned int) end_bit, stream));
    return MS_OK;
}

int sort_doubles_desc(void *temp, size_t *temp_bytes, const double *in, double *out, size_t n, hipStream_t stream) {
    MS_HIP(rocprim::radix_sort_keys_desc(temp, *temp_bytes, in, out, n, 0u, 64u, stream));
    return MS_OK;
}

int exclusive_sum_u32(void *temp, size_t *temp_bytes, const uint32_t *in, uint64_t *out, size_t n, hipStream_t stream) {
    MS_HIP(rocprim::exclusive_scan(temp, *temp_bytes, in, out, (uint64_t) 0, n, rocprim::plus<uint64_t>(), stream));
    return MS_OK;
}

}  // namespace ms
