// gather_rate.hip -- what a scattered table read costs the texture addressers by width: every lane of a wave reads entries of 4 / 8 / 16
// bytes at lane-random offsets inside a 430 KB table (the fp64 stage's shape: L2-resident, no reuse inside a wave), 16 reads in flight.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/gather_rate.hip -o tools/ubench/gather_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

// lines: how many distinct 64-byte lines the 64 lanes of a wave touch per read (64: every lane its own; 8: eight lanes share a line, each its own 16 / 8 / 4 bytes of it ...)
template <typename T>
__global__ void __launch_bounds__(256) gather(const char *__restrict__ tab, uint32_t mask, int iters, double *out, int lines) {
    const uint32_t lane = threadIdx.x & 63u, grp = lane % (uint32_t) lines;
    uint32_t s = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 64u * 2654435761u + grp * 40503u + 12345u;      // lanes of a group walk the same sequence
    double acc = 0;
    for (int it = 0; it < iters; it++) {
        T v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            s = s * 1664525u + 1013904223u;
            v[k] = *reinterpret_cast<const T *>(tab + (((s >> 8) & mask & ~63u) | ((lane / (uint32_t) lines * (uint32_t) sizeof(T)) & 63u)));
        }
#pragma unroll
        for (int k = 0; k < 16; k++) acc += (double) reinterpret_cast<const uint32_t *>(&v[k])[0];
    }
    if (acc == 1.2345) out[0] = acc;
}

template <typename T>
void run(const char *name, const char *tab, double *out, int lines) {
    const int iters = 64, blocks = 256 * 16;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    gather<T><<<blocks, 256>>>(tab, (1u << 19) - 1, iters, out, lines);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(a);
        gather<T><<<blocks, 256>>>(tab, (1u << 19) - 1, iters, out, lines);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double wave_reads = (double) blocks * 4 * iters * 16;
    printf("%2d lines per wave-read, %-28s %.3f ms  %.1f M wave-reads  -> %.2f ns per wave-read per CU = %.1f cycles at 2.1 GHz\n", lines, name, best, wave_reads / 1e6,
           best * 1e6 / (wave_reads / 256), best * 1e6 / (wave_reads / 256) * 2.1);
}

int main() {
    char *tab; double *out;
    hipMalloc(&tab, 1 << 20); hipMemset(tab, 1, 1 << 20); hipMalloc(&out, 64);
    for (int lines : {64, 32, 16, 8, 4, 1}) {
        run<uint32_t>("4 bytes per lane (dword)", tab, out, lines);
        run<uint2>("8 bytes per lane (dwordx2)", tab, out, lines);
        run<uint4>("16 bytes per lane (dwordx4)", tab, out, lines);
    }
    return 0;
}
