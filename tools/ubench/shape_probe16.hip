// shape_probe16.hip -- round 6, VERDICT r5 #1 (i): would the 16x16x128 shape of v_mfma_scale_f32_*_f8f6f4 run the pre-filter's dominant class
// (paired row tiles of two half-blocks: 59 % of the plan's fields) faster than the 32x32x64 shape it uses?
//
// Per trip a wave does what the product kernel does for ONE such row tile against 64 window starts:
//   form A (the product): the A operand from LDS (three ds_read_b128: 48 bytes per lane), four v_mfma_scale_f32_32x32x64_f8f6f4 (two chained pairs:
//           the second instruction of a pair reads the first one's 16 result registers back as C), the OR-chain over the 32 result registers,
//           one ballot;
//   form B: the same 48 bytes per lane (two 16-row operands of K = 128), EIGHT v_mfma_scale_f32_16x16x128_f8f6f4 (16 rows x 16 windows x K = 128
//           each, no chaining: C is the inline constant), the OR-chain over the same 32 result registers, one ballot.
// Both issue 128 matrix-pipe cycles per trip (4 x 32 = 8 x 16) and leave 32 registers to inspect; what form B saves is the accumulator read-back.
// 1024 threads per block = four waves per SIMD, one block per CU, as the product kernel.  Operands hold pseudo-random fp6 / one-hot-like fp4
// patterns (a quarter of the k-slots non-zero).  Reported: ns per trip and wave, matrix instructions per microsecond and SIMD, shader clock.
// Build: hipcc -O3 --offload-arch=gfx950 shape_probe16.hip -o shape_probe16.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) i32x4 lds_i32x4;

__device__ __forceinline__ unsigned or3(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xFE); }

template <int FORM, bool INSPECT>
__global__ void __launch_bounds__(1024, 4) probe(const i32x4 *tab, int n_tiles, int trips, unsigned long long *out, unsigned *sink, unsigned magic) {
    extern __shared__ i32x4 lds[];
    for (int i = threadIdx.x; i < n_tiles * 64 * 3; i += 1024) lds[i] = tab[i];
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u;
    i32x4 b[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const unsigned x = lane * 2654435761u + k * 40503u; b[k] = i32x4{(int) (0x2u << (4 * (x & 7))), (int) (0x2u << (4 * ((x >> 3) & 7))), (int) (0x2u << (4 * ((x >> 6) & 7))), (int) (0x2u << (4 * ((x >> 9) & 7)))}; }
    const int s0 = (lane >> 5) ? 109 : 121, s1 = s0 - 1;
    unsigned acc_or = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < trips; t++) {
        const lds_i32x4 *q = (const lds_i32x4 *) (lds + ((t % n_tiles) * 64 + lane) * 3);
        const i32x4 w0 = q[0], w1 = q[1], w2 = q[2];
        const i32x8 a0 = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], 0, 0}, a1 = {w1[2], w1[3], w2[0], w2[1], w2[2], w2[3], 0, 0};
        unsigned x;
        if constexpr (FORM == 0) {
            f32x16 cc0, cc1;
#pragma unroll
            for (int j = 0; j < 16; j++) { cc0[j] = 4.0f; cc1[j] = 2.0f; }
            f32x16 c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, i32x8{b[0][0], b[0][1], b[0][2], b[0][3], 0, 0, 0, 0}, cc0, 2, 4, 0, s0, 0, 127);
            f32x16 c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, i32x8{b[1][0], b[1][1], b[1][2], b[1][3], 0, 0, 0, 0}, cc1, 2, 4, 0, s1, 0, 127);
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, i32x8{b[2][0], b[2][1], b[2][2], b[2][3], 0, 0, 0, 0}, c0, 2, 4, 0, s0, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, i32x8{b[3][0], b[3][1], b[3][2], b[3][3], 0, 0, 0, 0}, c1, 2, 4, 0, s1, 0, 127);
            if constexpr (INSPECT) {
                unsigned y0 = or3(__float_as_uint(c0[0]), __float_as_uint(c0[1]), __float_as_uint(c0[2])), y1 = or3(__float_as_uint(c1[0]), __float_as_uint(c1[1]), __float_as_uint(c1[2]));
#pragma unroll
                for (int i = 3; i < 15; i += 2) { y0 = or3(y0, __float_as_uint(c0[i]), __float_as_uint(c0[i + 1])); y1 = or3(y1, __float_as_uint(c1[i]), __float_as_uint(c1[i + 1])); }
                x = y0 | y1 | __float_as_uint(c0[15]) | __float_as_uint(c1[15]);
            } else { asm volatile("" : : "v"(c0), "v"(c1)); x = 0; }
        } else {
            const f32x4 k4 = {4.0f, 4.0f, 4.0f, 4.0f}, k2 = {2.0f, 2.0f, 2.0f, 2.0f};
            f32x4 c[8];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const i32x8 bb = {b[g][0], b[g][1], b[g][2], b[g][3], 0, 0, 0, 0};
                c[2 * g] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a0, bb, (g & 1) ? k2 : k4, 2, 4, 0, (g & 1) ? s1 : s0, 0, 127);
                c[2 * g + 1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a1, bb, (g & 1) ? k2 : k4, 2, 4, 0, (g & 1) ? s1 : s0, 0, 127);
            }
            if constexpr (INSPECT) {
                unsigned y0 = or3(__float_as_uint(c[0][0]), __float_as_uint(c[0][1]), __float_as_uint(c[0][2])), y1 = or3(__float_as_uint(c[4][0]), __float_as_uint(c[4][1]), __float_as_uint(c[4][2]));
                y0 = or3(y0, __float_as_uint(c[0][3]), __float_as_uint(c[1][0])); y1 = or3(y1, __float_as_uint(c[4][3]), __float_as_uint(c[5][0]));
                y0 = or3(y0, __float_as_uint(c[1][1]), __float_as_uint(c[1][2])); y1 = or3(y1, __float_as_uint(c[5][1]), __float_as_uint(c[5][2]));
                y0 = or3(y0, __float_as_uint(c[1][3]), __float_as_uint(c[2][0])); y1 = or3(y1, __float_as_uint(c[5][3]), __float_as_uint(c[6][0]));
                y0 = or3(y0, __float_as_uint(c[2][1]), __float_as_uint(c[2][2])); y1 = or3(y1, __float_as_uint(c[6][1]), __float_as_uint(c[6][2]));
                y0 = or3(y0, __float_as_uint(c[2][3]), __float_as_uint(c[3][0])); y1 = or3(y1, __float_as_uint(c[6][3]), __float_as_uint(c[7][0]));
                y0 = or3(y0, __float_as_uint(c[3][1]), __float_as_uint(c[3][2])); y1 = or3(y1, __float_as_uint(c[7][1]), __float_as_uint(c[7][2]));
                x = y0 | y1 | __float_as_uint(c[3][3]) | __float_as_uint(c[7][3]);
            } else {
#pragma unroll
                for (int g = 0; g < 8; g++) asm volatile("" : : "v"(c[g]));
                x = 0;
            }
        }
        if constexpr (INSPECT) { if (__builtin_amdgcn_ballot_w64((x & magic) == 0x80000001u)) acc_or |= x; }        // (never true: the branch the product takes on an event)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (acc_or == 0xdeadbeefu) sink[0] = acc_or;
}

int main(int argc, char **argv) {
    const int trips = argc > 1 ? atoi(argv[1]) : 200000, n_tiles = 12, blocks = 256;
    i32x4 *tab; unsigned long long *out; unsigned *sink;
    CK(hipMalloc(&tab, n_tiles * 64 * 3 * sizeof(i32x4))); CK(hipMalloc(&out, blocks * 16)); CK(hipMalloc(&sink, 64));
    {
        unsigned *h = (unsigned *) malloc(n_tiles * 64 * 3 * 16);
        unsigned s = 12345u;
        for (int i = 0; i < n_tiles * 64 * 12; i++) { s = s * 1664525u + 1013904223u; h[i] = s & 0x3CF3CF3Cu; }      // fp6 codes of moderate magnitude
        CK(hipMemcpy(tab, h, n_tiles * 64 * 3 * 16, hipMemcpyHostToDevice)); free(h);
    }
    const size_t lds_bytes = n_tiles * 64 * 3 * 16;
    struct { const char *name; void (*k)(const i32x4 *, int, int, unsigned long long *, unsigned *, unsigned); int n_mfma; } forms[] = {
        {"A: 4 x 32x32x64 (chained pairs) + inspection", probe<0, true>, 4}, {"B: 8 x 16x16x128 (no chaining)  + inspection", probe<1, true>, 8},
        {"A without inspection", probe<0, false>, 4}, {"B without inspection", probe<1, false>, 8}};
    for (int rep = 0; rep < 3; rep++)
        for (auto &f : forms) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(f.k, dim3(blocks), dim3(1024), lds_bytes, 0, tab, n_tiles, trips / 10, out, sink, 0x00400400u); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(f.k, dim3(blocks), dim3(1024), lds_bytes, 0, tab, n_tiles, trips, out, sink, 0x00400400u); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[2 * 256]; CK(hipMemcpy(h, out, blocks * 16, hipMemcpyDeviceToHost));
            double cyc = 0, real = 0; for (int i = 0; i < blocks; i++) { cyc += (double) h[2 * i]; real += (double) h[2 * i + 1]; }
            const double mhz = cyc / real * 100.0;
            printf("%-48s %8.3f ms  %7.1f ns per trip and wave  %6.1f clk of the SIMD per trip (4 waves)  %5.1f matrix instr / us / SIMD  clock %4.0f MHz\n", f.name, ms,
                   ms * 1e6 / trips, ms * 1e-3 * mhz * 1e6 / trips / 4.0 * 1.0, f.n_mfma * 4.0 * trips / (ms * 1e3), mhz);
        }
    return 0;
}
