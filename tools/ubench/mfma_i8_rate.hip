// Micro-benchmark (measurement only): cycles per v_mfma_i32_32x32x32_i8 on one SIMD of an MI355X, for the
// issue patterns the engine-1 pre-filter uses.  Build: hipcc -O3 --offload-arch=gfx950 mfma_i8_rate.hip -o mfma_i8_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// MODE 0: 4 independent accumulators, accumulate in place (classic GEMM inner loop)
// MODE 1: pairs: c0 = mfma(a, b0, 0); c1 = mfma(a, b1, 0); c0 = mfma(a', b0', c0); c1 = mfma(a', b1', c1); consume one lane value
// MODE 2: like 1 but results are not consumed until the end (no VALU read of the accumulators in the loop)
template <int MODE>
__global__ void __launch_bounds__(1024) k(int iters, int *out, unsigned long long *cyc) {
    i32x4 a = {(int) threadIdx.x, 1, 2, 3}, a2 = {5, 6, 7, (int) threadIdx.x};
    i32x4 b0 = {1, 0x100, 0x10000, 1}, b1 = {0x100, 1, 1, 0x1000000};
    const i32x16 z = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    i32x16 c0 = z, c1 = z, c2 = z, c3 = z;
    int sink = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b1, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, b0, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, b1, c3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b0, z, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b1, z, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, b1, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, b0, c1, 0, 0, 0);
            if (MODE == 1) { sink |= c0[0] | c1[15]; a.x ^= sink & 1; }
            else { c2[0] += c0[3]; asm volatile("" :: "v"(c1[2])); a.x += 1; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int r = sink;
    for (int j = 0; j < 16; j++) r += c0[j] + c1[j] + c2[j] + c3[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) atomicMax(&cyc[blockIdx.x], t1 - t0);      // slowest wave of the block
}

__device__ __forceinline__ int max16(const i32x16 &c) {
    int a = max(max(c[0], c[1]), c[2]);
    int b = max(max(c[3], c[4]), c[5]);
    int d = max(max(c[6], c[7]), c[8]);
    int e = max(max(c[9], c[10]), c[11]);
    int f = max(max(c[12], c[13]), c[14]);
    a = max(max(a, b), d);
    e = max(max(e, f), c[15]);
    return max(a, e);
}

// MODE 3: the production tile: 2 ds_read_b128, 4 MFMA, 2 x max16, test, rarely-taken branch
// MODE 4: the same, tiles software-pipelined in pairs
// MODE 5: MODE 3 with only one max16 per tile (half the VALU work)
template <int MODE>
__global__ void __launch_bounds__(1024) kt(int iters, int *out, unsigned long long *cyc) {
    __shared__ uint4 tab[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = make_uint4(0x81828384u + i, 0x85868788u, 0x898a8b8cu, 0x8d8e8f80u);
    __syncthreads();
    const char *p = reinterpret_cast<const char *>(tab) + (threadIdx.x & 63) * 16;
    i32x4 b0 = {1, 0x100, 0x10000, 1}, b1 = {0x100, 1, 1, 0x1000000}, b2 = {0x10000, 1, 0x100, 1}, b3 = {1, 1, 0x1000000, 0x100};
    const i32x16 z = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    int sink = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    auto tile = [&](int t, i32x16 &c0, i32x16 &c1) {
        const char *q = p + (t & 31) * 2048;
        const i32x4 a0 = *reinterpret_cast<const i32x4 *>(q), a1 = *reinterpret_cast<const i32x4 *>(q + 1024);
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, z, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, z, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b2, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b3, c1, 0, 0, 0);
    };
    auto test = [&](const i32x16 &c0, const i32x16 &c1, int t) {
        const int m0 = max16(c0), m1 = MODE == 5 ? c1[3] : max16(c1);
        if (__any((m0 & m1) >= 0)) { sink += t; out[threadIdx.x] = t; }
    };
    if (MODE == 7) {
        // the same tile (32 rows x 64 windows x K = 64) from eight 16x16x64 instructions: 2 row halves x 4 window quarters
        typedef int i32x4v __attribute__((ext_vector_type(4)));
        const i32x4v z4 = {0, 0, 0, 0};
        for (int t = 0; t < iters; t++) {
            const char *q = p + (t & 31) * 2048;
            const i32x4 a0 = *reinterpret_cast<const i32x4 *>(q), a1 = *reinterpret_cast<const i32x4 *>(q + 1024);
            i32x4v c[8];
            c[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, z4, 0, 0, 0);
            c[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b1, z4, 0, 0, 0);
            c[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b2, z4, 0, 0, 0);
            c[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b3, z4, 0, 0, 0);
            c[4] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b0, z4, 0, 0, 0);
            c[5] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, z4, 0, 0, 0);
            c[6] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b2, z4, 0, 0, 0);
            c[7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b3, z4, 0, 0, 0);
            int m[11];
            for (int i = 0; i < 8; i++) m[i] = max(max(c[i][0], c[i][1]), c[i][2]);
            m[8] = max(max(c[0][3], c[1][3]), c[2][3]);
            m[9] = max(max(c[3][3], c[4][3]), c[5][3]);
            m[10] = max(c[6][3], c[7][3]);
            const int x = max(max(m[0], m[1]), m[2]), y = max(max(m[3], m[4]), m[5]), w = max(max(m[6], m[7]), m[8]);
            const int r = max(max(x, y), max(max(w, m[9]), m[10]));
            if (__any(r >= 0)) { sink += t; out[threadIdx.x] = t; }
        }
    } else if (MODE == 6) {
        // software pipeline with the A operands fetched one tile ahead: per iteration  read(t+2) | product(t+1) | reduce(t)
        auto rd = [&](int t, i32x4 &a0, i32x4 &a1) {
            const char *q = p + (t & 31) * 2048;
            a0 = *reinterpret_cast<const i32x4 *>(q); a1 = *reinterpret_cast<const i32x4 *>(q + 1024);
        };
        auto prod = [&](const i32x4 &a0, const i32x4 &a1, i32x16 &c0, i32x16 &c1) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, z, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, z, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b2, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b3, c1, 0, 0, 0);
        };
        i32x4 pa0, pa1, qa0, qa1;
        i32x16 xa0, xa1, xb0, xb1;
        rd(0, pa0, pa1);
        prod(pa0, pa1, xa0, xa1);
        rd(1, qa0, qa1);
        for (int t = 1; t + 1 < iters; t += 2) {
            rd(t + 1, pa0, pa1);
            prod(qa0, qa1, xb0, xb1);
            test(xa0, xa1, t - 1);
            rd(t + 2, qa0, qa1);
            prod(pa0, pa1, xa0, xa1);
            test(xb0, xb1, t);
        }
        test(xa0, xa1, iters);
    } else if (MODE == 4) {
        i32x16 xa0, xa1, xb0, xb1;
        tile(0, xa0, xa1);
        for (int t = 1; t + 1 < iters; t += 2) {
            tile(t, xb0, xb1);
            test(xa0, xa1, t - 1);
            tile(t + 1, xa0, xa1);
            test(xb0, xb1, t);
        }
        test(xa0, xa1, iters);
    } else {
        for (int t = 0; t < iters; t++) {
            i32x16 c0, c1;
            tile(t, c0, c1);
            test(c0, c1, t);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
    if ((threadIdx.x & 63) == 0) atomicMax(&cyc[blockIdx.x], t1 - t0);
}

template <int MODE> void run(const char *name, int threads, int blocks) {
    int *out; unsigned long long *cyc;
    hipMalloc(&out, sizeof(int) * threads * blocks); hipMalloc(&cyc, 8 * blocks);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipMemset(cyc, 0, 8 * blocks);
        hipEventRecord(e0, 0);
        if (MODE >= 3) hipLaunchKernelGGL(kt<MODE>, dim3(blocks), dim3(threads), 0, 0, iters, out, cyc); else hipLaunchKernelGGL(k<(MODE < 3 ? MODE : 0)>, dim3(blocks), dim3(threads), 0, 0, iters, out, cyc);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long h[1]; hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    const double waves_per_simd = threads / 256.0;
    printf("%-44s threads %4d blocks %3d: slowest wave %.1f cycles per MFMA per SIMD; wall %.3f ms = %.2f ns per MFMA per SIMD\n", name, threads, blocks,
           (double) h[0] / (iters * 4.0 * waves_per_simd), ms, ms * 1e6 / (iters * 4.0 * waves_per_simd));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int threads : {256, 512, 1024}) {
        for (int blocks : {256}) {
            run<0>("4 independent accumulating chains", threads, blocks);
            run<1>("zero-init pairs, result read by VALU", threads, blocks);
            run<2>("zero-init pairs, one register read", threads, blocks);
            run<3>("production tile (2 reads, 4 MFMA, 2 max16)", threads, blocks);
            run<4>("production tile, pipelined in pairs", threads, blocks);
            run<5>("production tile, one max16 only", threads, blocks);
            run<6>("production tile, pipelined + operand prefetch", threads, blocks);
            run<7>("production tile from 8 x 16x16x64", threads, blocks);
        }
    }
    return 0;
}
