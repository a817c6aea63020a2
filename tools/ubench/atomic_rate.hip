// atomic_rate.hip -- measurement only: how many RETURNING global atomicAdds per second the chip sustains when the words are many
// (a bucket-fill table of 4 k ... 1 M counters, as a direct placement of hits into per-(motif, region range) buckets would use) and each
// wave issues them from a few lanes at a time.  Round 4: decides whether hits can be placed at emission instead of being sorted.
//   ./atomic_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// every thread: n_per_thread atomics on pseudo-random words of a table of n_words; lanes_active of each wave's 64 lanes take part
__global__ void k(unsigned int *tab, unsigned int n_words, int n_per_thread, int lanes_active, unsigned int *sink) {
    const unsigned int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if ((int) (threadIdx.x & 63u) >= lanes_active) return;
    unsigned int x = tid * 2654435761u + 12345u, acc = 0;
    for (int i = 0; i < n_per_thread; i++) {
        x = x * 1664525u + 1013904223u;
        acc += atomicAdd(&tab[(x >> 8) % n_words], 1u);
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

int main() {
    unsigned int *tab, *sink;
    CK(hipMalloc(&tab, (1u << 20) * 4));
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 4, threads = 256;
    for (int lanes : {64, 8, 1})
        for (unsigned int n_words : {579u, 4096u, 40000u, 250000u, 1u << 20}) {
            const int per = lanes == 64 ? 64 : (lanes == 8 ? 256 : 1024);
            CK(hipMemset(tab, 0, (1u << 20) * 4));
            k<<<blocks, threads>>>(tab, n_words, 8, lanes, sink);
            CK(hipEventRecord(e0));
            k<<<blocks, threads>>>(tab, n_words, per, lanes, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double n = (double) blocks * threads / 64 * lanes * per;
            printf("%2d lanes per wave, %7u words: %.1f M atomics in %.3f ms = %.2f G/s\n", lanes, n_words, n / 1e6, ms, n / ms / 1e6);
        }
    return 0;
}
