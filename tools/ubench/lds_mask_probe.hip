// lds_mask_probe.hip -- measurement only (gfx950): what does an LDS store cost the CU's LDS pipe when only a few of the wave's lanes
// are active?  The pre-filter's park event stores 4 x ds_write_b128 + 1 x ds_write_b64 under the candidate lanes' exec mask
// (usually ONE lane); SQ_LDS_IDX_ACTIVE says the pipe is ~70 % busy in that kernel, so the question is whether those five
// instructions take the pipe for 36 cycles or for ~5.
//   per configuration: 16 waves per CU, every wave loops over {4 x b128 + 1 x b64} stores into its own LDS slice with
//   n active lanes (1, 2, 4, 16 = one quarter, 32, 64); reported: cycles per event per CU (all 16 waves issue back to back, so the
//   figure is the pipe's occupancy per event when it is the bound, the issue cost otherwise), and the same for READS (ds_read_b128).
// Build: hipcc -O2 --offload-arch=gfx950 lds_mask_probe.hip -o lds_mask_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <bool READ>
__global__ void __launch_bounds__(1024) probe(int trips, int n_active, int spread, unsigned int *sink, unsigned long long *clk) {
    __shared__ u32x4 space[16 * 64 * 5];                       // 80 KB: 5 x 16 B per lane and wave
    const unsigned int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // active lanes: the first n_active (spread = 0) or every (64 / n_active)-th (spread = 1)
    const bool on = spread ? (lane % (64 / n_active) == 0) : ((int) lane < n_active);
    u32x4 *mine = space + (wave * 64 + lane) * 5;
    u32x4 v = {lane, wave, lane * 3u, 7u};
    unsigned int acc = 0;
    for (int i = threadIdx.x; i < 16 * 64 * 5; i += blockDim.x) space[i] = v;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (on) {
        for (int t = 0; t < trips; t++) {
            if constexpr (READ) {
                u32x4 a = mine[0], b = mine[1], c = mine[2], d = mine[3];
                u32x2 e = *reinterpret_cast<u32x2 *>(mine + 4);
                asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d), "v"(e));
            } else {
                asm volatile("" : "+v"(v));
                mine[0] = v; mine[1] = v; mine[2] = v; mine[3] = v;
                *reinterpret_cast<u32x2 *>(mine + 4) = u32x2{v.x, v.y};
                asm volatile("" ::: "memory");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    acc = space[threadIdx.x].x;
    if (acc == 0x12345u) sink[0] = acc;
    if (lane == 0) { clk[2 * (blockIdx.x * 16 + wave)] = t1 - t0; clk[2 * (blockIdx.x * 16 + wave) + 1] = r1 - r0; }
}

// the same event through VECTOR-MEMORY stores into a per-wave ring in global memory (L2-resident: 3.6 KB per wave): 4 x global_store_dwordx4
// + 1 x global_store_dwordx2 under the mask; every 48th event the wave waits for its stores (s_waitcnt vmcnt(0)) as a flush would
__global__ void __launch_bounds__(1024) vprobe(int trips, int n_active, int spread, u32x4 *ring, unsigned long long *clk) {
    const unsigned int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool on = spread ? (lane % (64 / n_active) == 0) : ((int) lane < n_active);
    u32x4 *mine = ring + ((size_t) (blockIdx.x * 16 + wave) * 48) * 5;          // 48 entries of 80 bytes per wave
    u32x4 v = {lane, wave, lane * 3u, 7u};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (on) {
        for (int t = 0; t < trips; t++) {
            u32x4 *e = mine + (size_t) ((t % 48) * 5);                            // one entry per event (the lanes of an event share it here:
            asm volatile("" : "+v"(v));                                          //  the address pattern is not the question, the issue cost is)
            e[0] = v; e[1] = v; e[2] = v; e[3] = v;
            *reinterpret_cast<u32x2 *>(e + 4) = u32x2{v.x, v.y};
            if (t % 48 == 47) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { clk[2 * (blockIdx.x * 16 + wave)] = t1 - t0; clk[2 * (blockIdx.x * 16 + wave) + 1] = 0; }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned int *d_sink; unsigned long long *d_clk;
    CK(hipMalloc(&d_sink, 64));
    CK(hipMalloc(&d_clk, 256 * 16 * 2 * 8));
    const int trips = 20000;
    for (int rd = 0; rd < 2; rd++) {
        printf("%s: 4 x b128 + 1 x b64 per event, 16 waves per CU, 256 CUs\n", rd ? "ds_read" : "ds_write");
        for (int spread = 0; spread < 2; spread++)
            for (int n : {1, 2, 4, 16, 32, 64}) {
                if (spread && (n == 64)) continue;
                for (int w = 0; w < 2; w++) {
                    if (rd) hipLaunchKernelGGL((probe<true>), dim3(256), dim3(1024), 0, 0, trips, n, spread, d_sink, d_clk);
                    else hipLaunchKernelGGL((probe<false>), dim3(256), dim3(1024), 0, 0, trips, n, spread, d_sink, d_clk);
                }
                CK(hipDeviceSynchronize());
                std::vector<unsigned long long> h(256 * 16 * 2);
                CK(hipMemcpy(h.data(), d_clk, h.size() * 8, hipMemcpyDeviceToHost));
                double cyc = 0; int cnt = 0;
                for (int i = 0; i < 256 * 16; i++) { cyc += (double) h[2 * i]; cnt++; }
                cyc /= cnt;
                // all 16 waves of the CU run the loop side by side: the CU completes 16 events in (cyc / trips) cycles
                printf("    %2d active lanes (%s): %7.1f cycles per event and wave = %6.1f cycles of the CU's LDS pipe per event\n",
                       n, spread ? "spread over the wave" : "the first lanes    ", cyc / trips, cyc / trips / 16.0);
            }
    }
    {
        u32x4 *d_ring;
        CK(hipMalloc(&d_ring, (size_t) 256 * 16 * 48 * 80));
        printf("global_store: 4 x dwordx4 + 1 x dwordx2 per event into a per-wave ring, 16 waves per CU, 256 CUs\n");
        for (int spread = 0; spread < 2; spread++)
            for (int n : {1, 2, 4, 16, 64}) {
                if (spread && n == 64) continue;
                for (int w = 0; w < 2; w++) hipLaunchKernelGGL(vprobe, dim3(256), dim3(1024), 0, 0, trips, n, spread, d_ring, d_clk);
                CK(hipDeviceSynchronize());
                std::vector<unsigned long long> h(256 * 16 * 2);
                CK(hipMemcpy(h.data(), d_clk, h.size() * 8, hipMemcpyDeviceToHost));
                double cyc = 0;
                for (int i = 0; i < 256 * 16; i++) cyc += (double) h[2 * i];
                cyc /= 256 * 16;
                printf("    %2d active lanes (%s): %7.1f cycles per event and wave = %6.1f cycles of the CU per event\n",
                       n, spread ? "spread over the wave" : "the first lanes    ", cyc / trips, cyc / trips / 16.0);
            }
    }
    return 0;
}
