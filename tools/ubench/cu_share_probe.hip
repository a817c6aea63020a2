// cu_share_probe.hip -- does a persistent kernel that fills (n_cu - r) CUs leave the other r CUs to a second stream's kernel?
// hog: 1024 threads x 128 VGPRs (one block per CU), spins T ms.  side: a short many-workgroup kernel (256 threads, streaming
// read of 64 MB) launched on another stream right after the hog.  Reported: the side kernel's duration alone and beside the hog
// for r = 0, 8, 16 (plain streams), and for r = 8 with the two streams confined by CU masks (hipExtStreamCreateWithCUMask).
// Build: hipcc -O2 --offload-arch=gfx950 cu_share_probe.hip -o cu_share_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

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

__global__ void __launch_bounds__(256) side_kernel(const uint4 *__restrict__ in, unsigned int *__restrict__ out, size_t n16) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const uint4 a = in[i];
    out[i] = a.x ^ a.y ^ a.z ^ a.w;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const size_t bytes = 64u << 20, n16 = bytes / 16;
    uint4 *d_in;
    unsigned int *d_out;
    float *d_sink;
    CK(hipMalloc(&d_in, bytes));
    CK(hipMalloc(&d_out, n16 * 4));
    CK(hipMalloc(&d_sink, 64));
    CK(hipMemset(d_in, 1, bytes));
    const long long ticks = 100000LL * 5;           // 5 ms at 100 MHz
    hipEvent_t e0, e1, h0, h1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&h0)); CK(hipEventCreate(&h1));
    auto run = [&](hipStream_t s_hog, hipStream_t s_side, int hog_blocks, const char *what) {
        for (int rep = 0; rep < 3; rep++) {
            if (hog_blocks) {
                CK(hipEventRecord(h0, s_hog));
                hipLaunchKernelGGL(hog_kernel, dim3((unsigned) hog_blocks), dim3(1024), 0, s_hog, d_sink, ticks);
                CK(hipEventRecord(h1, s_hog));
            }
            CK(hipEventRecord(e0, s_side));
            hipLaunchKernelGGL(side_kernel, dim3((unsigned) ((n16 + 255) / 256)), dim3(256), 0, s_side, d_in, d_out, n16);
            CK(hipEventRecord(e1, s_side));
            CK(hipDeviceSynchronize());
            float ms = 0, hms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (hog_blocks) CK(hipEventElapsedTime(&hms, h0, h1));
            if (rep == 2) printf("%-58s side kernel %7.3f ms   hog %6.3f ms\n", what, ms, hms);
        }
    };
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    run(a, b, 0, "alone");
    char buf[128];
    for (int r : {0, 8, 16, 32}) {
        snprintf(buf, sizeof buf, "beside a hog of %d blocks (plain streams)", n_cu - r);
        run(a, b, n_cu - r, buf);
    }
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t bp;
    CK(hipStreamCreateWithPriority(&bp, hipStreamNonBlocking, greatest));
    for (int r : {0, 8, 16}) {
        snprintf(buf, sizeof buf, "beside a hog of %d blocks (side stream high priority)", n_cu - r);
        run(a, bp, n_cu - r, buf);
    }
    for (int r : {8, 16}) {
        std::vector<uint32_t> m_side((size_t) (n_cu + 31) / 32, 0u), m_hog((size_t) (n_cu + 31) / 32, 0u);
        for (int i = 0; i < n_cu; i++) (i < r ? m_side : m_hog)[(size_t) i / 32] |= 1u << (i % 32);
        hipStream_t ms_, mh;
        CK(hipExtStreamCreateWithCUMask(&ms_, (uint32_t) m_side.size(), m_side.data()));
        CK(hipExtStreamCreateWithCUMask(&mh, (uint32_t) m_hog.size(), m_hog.data()));
        snprintf(buf, sizeof buf, "alone on a stream masked to %d CUs", r);
        run(mh, ms_, 0, buf);
        snprintf(buf, sizeof buf, "alone on a stream masked to the OTHER %d CUs", n_cu - r);
        run(ms_, mh, 0, buf);
        // a long many-workgroup kernel (16 x the side kernel back to back) on the big share vs unmasked
        for (int which = 0; which < 2; which++) {
            hipStream_t st = which ? mh : a;
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < 16; k++) hipLaunchKernelGGL(side_kernel, dim3((unsigned) ((n16 + 255) / 256)), dim3(256), 0, st, d_in, d_out, n16);
            CK(hipEventRecord(e1, st));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("16 side kernels back to back, %s: %.3f ms\n", which ? "masked to the big share" : "unmasked", ms);
        }
        snprintf(buf, sizeof buf, "hog of %d blocks masked to the other CUs, side masked to %d", n_cu - r, r);
        run(mh, ms_, n_cu - r, buf);
        snprintf(buf, sizeof buf, "hog of %d blocks UNMASKED, side masked to %d CUs", n_cu - r, r);
        run(a, ms_, n_cu - r, buf);
    }
    return 0;
}
