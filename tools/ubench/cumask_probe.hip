// cumask_probe.hip -- can the copy streams be confined to a few CUs and still move data at the link rate beside a kernel that
// fills every other CU?  (hipExtStreamCreateWithCUMask)
//   1. which CUs a mask selects: a marker kernel records (XCC_ID, HW_ID[15:8]) per workgroup, distinct keys are counted;
//   2. H2D / D2H rate of hipMemcpyAsync (pinned host memory, 128 MB) on a stream masked to k CUs, alone;
//   3. the same while a register-hungry persistent kernel (1024 threads, 128 VGPRs: one block fills a CU, like the pre-filter)
//      occupies the complement of the mask -- and, for contrast, while it occupies ALL CUs with the copy stream unmasked.
// Build: hipcc -O2 --offload-arch=gfx950 cumask_probe.hip -o cumask_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void where_kernel(unsigned int *out) {
    if (threadIdx.x == 0) {
        unsigned int hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = ((xcc & 0xFu) << 8) | ((hw >> 8) & 0xFFu);
    }
    // stay a little so that the workgroups spread over everything the mask allows
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}
}

// one block fills a CU: 1024 threads x 128 VGPRs
__global__ void __launch_bounds__(1024, 4) hog_kernel(float *sink, long long ticks) {
    float v[96];
#pragma unroll
    for (int i = 0; i < 96; i++) v[i] = (float) (threadIdx.x + i);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 96; i++) v[i] = v[i] * 1.0001f + v[(i + 1) % 96];
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 96; i++) s += v[i];
    if (s == 12345.678f) sink[0] = s;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static std::vector<uint32_t> make_mask(int n_cu, const std::vector<int> &cus) {
    std::vector<uint32_t> m((size_t) (n_cu + 31) / 32, 0u);
    for (int c : cus) m[(size_t) c / 32] |= 1u << (c % 32);
    return m;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs, wall clock %d kHz\n", prop.name, n_cu, prop.clockRate);
    const size_t bytes = 128u << 20;
    void *h = nullptr, *d = nullptr;
    CK(hipHostMalloc(&h, bytes));
    CK(hipMalloc(&d, bytes));
    memset(h, 1, bytes);
    unsigned int *d_where;
    float *d_sink;
    const int n_wg = 8192;
    CK(hipMalloc(&d_where, n_wg * sizeof(unsigned int)));
    CK(hipMalloc(&d_sink, 64));
    std::vector<unsigned int> where(n_wg);
    const long long hog_ticks = 100000000LL * 30 / 1000;     // ~30 ms at 100 MHz wall clock

    for (int k : {4, 8, 16, 32}) {
        for (int layout = 0; layout < 2; layout++) {
            // layout 0: the first k CU bits; layout 1: every (n_cu / k)-th bit
            std::vector<int> cus, rest;
            std::vector<char> in((size_t) n_cu, 0);
            for (int i = 0; i < k; i++) in[(size_t) (layout ? i * (n_cu / k) : i)] = 1;
            for (int i = 0; i < n_cu; i++) (in[(size_t) i] ? cus : rest).push_back(i);
            auto m_copy = make_mask(n_cu, cus), m_rest = make_mask(n_cu, rest);
            hipStream_t s_copy, s_rest, s_all;
            CK(hipExtStreamCreateWithCUMask(&s_copy, (uint32_t) m_copy.size(), m_copy.data()));
            CK(hipExtStreamCreateWithCUMask(&s_rest, (uint32_t) m_rest.size(), m_rest.data()));
            CK(hipStreamCreateWithFlags(&s_all, hipStreamNonBlocking));
            // 1. where do the masked streams run?
            auto distinct = [&](hipStream_t s) {
                hipLaunchKernelGGL(where_kernel, dim3(n_wg), dim3(64), 0, s, d_where);
                CK(hipStreamSynchronize(s));
                CK(hipMemcpy(where.data(), d_where, n_wg * sizeof(unsigned int), hipMemcpyDeviceToHost));
                return std::set<unsigned int>(where.begin(), where.end());
            };
            auto set_copy = distinct(s_copy), set_rest = distinct(s_rest);
            size_t overlap = 0;
            for (unsigned int x : set_copy) overlap += set_rest.count(x);
            std::set<unsigned int> xccs;
            for (unsigned int x : set_copy) xccs.insert(x >> 8);
            printf("k=%2d layout %d: copy mask -> %zu distinct CUs on %zu XCCs, complement -> %zu, overlap %zu\n", k, layout,
                   set_copy.size(), xccs.size(), set_rest.size(), overlap);
            // 2./3. copy rates
            auto rate = [&](hipStream_t s, int dir) {
                double best = 1e9;
                for (int rep = 0; rep < 5; rep++) {
                    const double t0 = now();
                    if (dir == 0) CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
                    else CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s));
                    CK(hipStreamSynchronize(s));
                    best = std::min(best, now() - t0);
                }
                return bytes / best / 1e9;
            };
            printf("    alone:                      H2D %6.1f GB/s  D2H %6.1f GB/s\n", rate(s_copy, 0), rate(s_copy, 1));
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(hog_kernel, dim3((unsigned) rest.size()), dim3(1024), 0, s_rest, d_sink, hog_ticks * 8);
                const double a = rate(s_copy, 0), b = rate(s_copy, 1);
                const hipError_t q = hipStreamQuery(s_rest);
                CK(hipStreamSynchronize(s_rest));
                printf("    beside a hog on the rest:   H2D %6.1f GB/s  D2H %6.1f GB/s   (hog still running: %s)\n", a, b,
                       q == hipErrorNotReady ? "yes" : "NO");
            }
            if (layout == 0 && k == 8) {
                hipLaunchKernelGGL(hog_kernel, dim3((unsigned) n_cu), dim3(1024), 0, s_all, d_sink, hog_ticks * 2);
                const double t0 = now();
                CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s_copy));
                CK(hipStreamSynchronize(s_copy));
                const double t1 = now();
                CK(hipStreamSynchronize(s_all));
                printf("    hog on ALL CUs (60 ms), masked copy of 128 MB took %.1f ms\n", (t1 - t0) * 1e3);
            }
            CK(hipStreamDestroy(s_copy));
            CK(hipStreamDestroy(s_rest));
            CK(hipStreamDestroy(s_all));
        }
    }
    return 0;
}
