// insp_probe.hip -- measurement only (gfx950): how to ask "is any of the 32 f32 results of a row tile >= +0" cheaply.
//   (1) layout of v_cvt_scalef32_2xpk16_fp6_f32 (32 f32 in two 16-register operands -> 32 fp6 in 6 registers): which input's
//       sign lands on which output bit, and that the sign survives for the values the pre-filter produces
//       (multiples of 1/8, |x| <= 7.5 ... and beyond: saturation).
//   (2) issue cost beside the matrix pipe: per trip 2 x v_mfma_scale_f32_32x32x64_f8f6f4 (C = 0) and one inspection of the
//       32 result registers, with 1 / 2 / 4 waves per SIMD and as two 512-thread blocks per CU:
//         mode 0  results only consumed (matrix pipe alone)
//         mode 1  16 x v_max3_i32 + compare + branch                     (round 2's kernel)
//         mode 2  cvt-pack + 3 v_and_or/and per half + and + compare + branch
//         mode 3  cvt-pack alone
//         mode 4 / 5  16 x v_bitop3_b32 (3-input AND of the sign bits) as a chain / as a tree + compare + branch
// Reported per configuration: wall time, shader clock (s_memtime against the 100 MHz s_memrealtime), and cycles per trip per
// SIMD = wall x clock / (trips x waves per SIMD) -- from the wall clock, not from one wave's own cycle count (the oldest wave
// of a SIMD wins the arbitration and finishes early: its count understates the SIMD's time).
// Build: hipcc -O3 --offload-arch=gfx950 insp_probe.hip -o insp_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x6 __attribute__((ext_vector_type(6)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void layout_kernel(const float *in, unsigned int *out) {          // in: [case][32], out: [case][6]
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    f32x16 a, b;
    for (int j = 0; j < 16; j++) { a[j] = in[t * 32 + j]; b[j] = in[t * 32 + 16 + j]; }
    const u32x6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, 1.0f);
    for (int k = 0; k < 6; k++) out[t * 6 + k] = r[k];
}

constexpr unsigned int kM0 = (1u << 5) | (1u << 11) | (1u << 17) | (1u << 23) | (1u << 29);
constexpr unsigned int kM1 = (1u << 3) | (1u << 9) | (1u << 15) | (1u << 21) | (1u << 27);
constexpr unsigned int kM2 = (1u << 1) | (1u << 7) | (1u << 13) | (1u << 19) | (1u << 25) | (1u << 31);

// modes 6 / 7: the AND-chain inspection of mode 4 with the A operand READ FROM LDS every trip, as the kernel does (three ds_read_b64 per
// row tile): 6 = read, wait, multiply (the kernel's order); 7 = the next trip's operand is read before this trip's inspection
template <int MODE>
__global__ void __launch_bounds__(1024) lds_kernel(const i32x8 *ab, int trips, unsigned int *sink, unsigned long long *clk) {
    __shared__ unsigned long long tab[32 * 3 * 64];                       // 32 row tiles of one k-block: [tile][plane][lane]
    for (int i = threadIdx.x; i < 32 * 3 * 64; i += blockDim.x) {
        const i32x8 v = ab[i & 63];
        const int pl = (i >> 6) % 3;
        tab[i] = ((unsigned long long) (unsigned int) v[2 * pl + 1] << 32) | (unsigned int) v[2 * pl];
    }
    __syncthreads();
    i32x8 b0 = ab[64 + (threadIdx.x & 63)], b1 = ab[128 + (threadIdx.x & 63)];
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned int found = 0;
    const unsigned int lane = threadIdx.x & 63;
    auto load = [&](int t) {
        const unsigned long long *q = tab + (t & 31) * 192 + lane;
        const unsigned long long w0 = q[0], w1 = q[64], w2 = q[128];
        return i32x8{(int) w0, (int) (w0 >> 32), (int) w1, (int) (w1 >> 32), (int) w2, (int) (w2 >> 32), 0, 0};
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    i32x8 a_next = load(0);
    for (int t = 0; t < trips; t++) {
        asm volatile("" : "+v"(b0), "+v"(b1));
        i32x8 a;
        if constexpr (MODE == 6) a = load(t); else { a = a_next; }
        const f32x16 c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b0, z, 2, 4, 0, 127, 0, 127);
        const f32x16 c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b1, z, 2, 4, 0, 127, 0, 127);
        if constexpr (MODE == 7) { a_next = load(t + 1); __builtin_amdgcn_sched_barrier(0); }
        unsigned int x = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 16; i++) x &= (unsigned int) __float_as_int(c0[i]) & (unsigned int) __float_as_int(c1[i]);
        if (__builtin_expect(__any((int) x >= 0), 0)) found += x;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (found == 0x12345u) sink[0] = found;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
__global__ void __launch_bounds__(1024) cost_kernel(const i32x8 *ab, int trips, unsigned int *sink, unsigned long long *clk) {
    i32x8 a = ab[threadIdx.x & 63], b0 = ab[64 + (threadIdx.x & 63)], b1 = ab[128 + (threadIdx.x & 63)];
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned int found = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < trips; t++) {
        asm volatile("" : "+v"(a), "+v"(b0), "+v"(b1));                   // opaque: the products are not loop-invariant for the compiler
        const f32x16 c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b0, z, 2, 4, 0, 127, 0, 127);
        const f32x16 c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b1, z, 2, 4, 0, 127, 0, 127);
        if constexpr (MODE == 0) {
            asm volatile("" :: "v"(c0), "v"(c1));
        } else if constexpr (MODE == 1) {
            int m[11];
#pragma unroll
            for (int i = 0; i < 5; i++) {
                m[i] = max(max(__float_as_int(c0[3 * i]), __float_as_int(c0[3 * i + 1])), __float_as_int(c0[3 * i + 2]));
                m[5 + i] = max(max(__float_as_int(c1[3 * i]), __float_as_int(c1[3 * i + 1])), __float_as_int(c1[3 * i + 2]));
            }
            m[10] = max(max(__float_as_int(c0[15]), __float_as_int(c1[15])), m[0]);
            const int x = max(max(m[1], m[2]), m[3]), y = max(max(m[4], m[5]), m[6]), zz = max(max(m[7], m[8]), m[9]);
            const int mx = max(max(x, y), max(zz, m[10]));
            if (__builtin_expect(__any(mx >= 0), 0)) found += (unsigned int) mx + 1u;
        } else if constexpr (MODE == 2) {
            const u32x6 p = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(c0, c1, 1.0f);
            const unsigned int f0 = (p[0] & kM0) | ((p[1] & kM1) | (p[2] & kM2));
            const unsigned int f1 = (p[3] & kM0) | ((p[4] & kM1) | (p[5] & kM2));
            const unsigned int fa = f0 & f1;
            if (__builtin_expect(__any(fa != 0xAAAAAAAAu), 0)) found += f0 ^ f1;
        } else if constexpr (MODE == 3) {
            const u32x6 p = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(c0, c1, 1.0f);
            asm volatile("" :: "v"(p));
        } else if constexpr (MODE == 4) {
            unsigned int x = 0xFFFFFFFFu;
#pragma unroll
            for (int i = 0; i < 16; i++) x &= (unsigned int) __float_as_int(c0[i]) & (unsigned int) __float_as_int(c1[i]);
            if (__builtin_expect(__any((int) x >= 0), 0)) found += x;
        } else {
            unsigned int m[11];                                                // the max3 tree of mode 1 with 3-input ANDs (v_bitop3_b32)
#pragma unroll
            for (int i = 0; i < 5; i++) {
                m[i] = (unsigned int) __float_as_int(c0[3 * i]) & (unsigned int) __float_as_int(c0[3 * i + 1]) & (unsigned int) __float_as_int(c0[3 * i + 2]);
                m[5 + i] = (unsigned int) __float_as_int(c1[3 * i]) & (unsigned int) __float_as_int(c1[3 * i + 1]) & (unsigned int) __float_as_int(c1[3 * i + 2]);
            }
            m[10] = (unsigned int) __float_as_int(c0[15]) & (unsigned int) __float_as_int(c1[15]) & m[0];
            const unsigned int x = m[1] & m[2] & m[3], y = m[4] & m[5] & m[6], zz = m[7] & m[8] & m[9];
            const unsigned int mx = (x & y & zz) & m[10];
            if (__builtin_expect(__any((int) mx >= 0), 0)) found += mx;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (found == 0x12345u) sink[0] = found;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

static void put_bits(unsigned *w, int bit, int nbits, unsigned v) {
    for (int i = 0; i < nbits; i++) if ((v >> i) & 1) w[(bit + i) >> 5] |= 1u << ((bit + i) & 31);
}

template <int MODE>
static void run_cost(const char *what, const i32x8 *d_ab, unsigned int *d_sink, unsigned long long *d_clk) {
    const int trips = 40000;
    printf("%-52s\n", what);
    struct Cfg { int blocks, threads; const char *name; } cfgs[] = {
        {256, 256, "1 wave/SIMD "}, {256, 512, "2 waves/SIMD"}, {256, 1024, "4 waves/SIMD"}, {512, 512, "2 x 512 per CU (4 waves/SIMD)"}};
    for (const Cfg &c : cfgs) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto launch = [&]() {
            if constexpr (MODE >= 6) hipLaunchKernelGGL((lds_kernel<MODE>), dim3(c.blocks), dim3(c.threads), 0, 0, d_ab, trips, d_sink, d_clk);
            else hipLaunchKernelGGL((cost_kernel<MODE>), dim3(c.blocks), dim3(c.threads), 0, 0, d_ab, trips, d_sink, d_clk);
        };
        for (int w = 0; w < 3; w++) launch();
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(2 * c.blocks);
        CK(hipMemcpy(h.data(), d_clk, h.size() * 8, hipMemcpyDeviceToHost));
        double mhz = 0; int n = 0;
        for (int b = 0; b < c.blocks; b++) if (h[2 * b + 1]) { mhz += 100.0 * (double) h[2 * b] / (double) h[2 * b + 1]; n++; }
        mhz /= n ? n : 1;
        const int wps = c.blocks * c.threads / 64 / 1024;                     // waves per SIMD (256 CUs x 4 SIMDs)
        const double cyc = ms * 1e-3 * mhz * 1e6 / ((double) trips * wps);
        printf("    %-32s %8.3f ms  clock %6.0f MHz  %6.1f cycles per trip per SIMD (2 MFMA + inspection)\n", c.name, ms, mhz, cyc);
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    // ---- (1) layout: case t = 32 inputs all +1.0 except input t = -1.0; then value cases ----
    {
        const int n_cases = 64;
        std::vector<float> in(n_cases * 32, 1.0f);
        for (int t = 0; t < 32; t++) in[t * 32 + t] = -1.0f;
        const float vals[8] = {-0.125f, 0.0f, -7.5f, -8.0f, -100.0f, 0.125f, 7.5f, 100.0f};
        for (int t = 32; t < 40; t++) for (int j = 0; j < 32; j++) in[t * 32 + j] = vals[t - 32];
        for (int t = 40; t < 64; t++) for (int j = 0; j < 32; j++) in[t * 32 + j] = -0.125f * (float) ((t - 40) * 32 + j + 1);   // -1/8 ... -96
        float *d_in; unsigned int *d_out;
        CK(hipMalloc(&d_in, in.size() * 4)); CK(hipMalloc(&d_out, n_cases * 6 * 4));
        CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(n_cases), 0, 0, d_in, d_out);
        std::vector<unsigned int> out(n_cases * 6);
        CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
        // reference: all +1.0 -> code 0b001000 everywhere; a -1.0 adds the sign bit 0b100000 somewhere
        unsigned int base[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 32; i++) put_bits(base, 6 * i, 6, 0x08);
        int hyp_ok = 0;
        for (int t = 0; t < 32; t++) {
            int where = -1, nflip = 0;
            for (int bit = 0; bit < 192; bit++) {
                const unsigned a = (out[t * 6 + (bit >> 5)] >> (bit & 31)) & 1u, b = (base[bit >> 5] >> (bit & 31)) & 1u;
                if (a != b) { where = bit; nflip++; }
            }
            printf("input %2d (operand %d, element %2d) = -1.0: %d bit(s) differ from all-(+1.0), bit %3d = value slot %2d, bit-in-slot %d\n",
                   t, t >> 4, t & 15, nflip, where, where / 6, where % 6);
            if (nflip == 1 && where == 6 * t + 5) hyp_ok++;
        }
        printf("hypothesis 'operand 0 element i -> bits [6i, 6i+6), operand 1 element i -> bits [96 + 6i, ...)': %s (%d of 32)\n",
               hyp_ok == 32 ? "CONFIRMED" : "NOT confirmed", hyp_ok);
        for (int t = 32; t < 40; t++) {
            const unsigned code = out[t * 6] & 63u;
            printf("value %8.3f -> fp6 code 0x%02x (sign %u)\n", vals[t - 32], code, code >> 5);
        }
        int sign_lost = 0;
        for (int t = 40; t < 64; t++)
            for (int j = 0; j < 32; j++) {
                const int bit = 6 * j + 5;
                if (!((out[t * 6 + (bit >> 5)] >> (bit & 31)) & 1u)) sign_lost++;
            }
        printf("negative multiples of 1/8 from -0.125 to -96: sign bit lost in %d of 768 conversions\n", sign_lost);
    }
    // ---- (2) cost ----
    std::vector<unsigned> ab(3 * 64 * 8, 0);
    for (int l = 0; l < 64; l++) {
        for (int j = 0; j < 32; j++) put_bits(&ab[l * 8], 6 * j, 6, 0x20u | (unsigned) (1 + ((l + j) % 24)));      // A: negative fp6 values
        for (int j = 0; j < 32; j++) put_bits(&ab[(64 + l) * 8], 4 * j, 4, ((j + l) & 3) == 0 ? 0x2u : 0u);       // B: one-hot fp4
        for (int j = 0; j < 32; j++) put_bits(&ab[(128 + l) * 8], 4 * j, 4, ((j + l) & 3) == 1 ? 0x2u : 0u);
    }
    i32x8 *d_ab; unsigned int *d_sink; unsigned long long *d_clk;
    CK(hipMalloc(&d_ab, ab.size() * 4)); CK(hipMalloc(&d_sink, 64)); CK(hipMalloc(&d_clk, 2 * 512 * 8));
    CK(hipMemcpy(d_ab, ab.data(), ab.size() * 4, hipMemcpyHostToDevice));
    run_cost<0>("mode 0: 2 MFMA, results only consumed", d_ab, d_sink, d_clk);
    run_cost<1>("mode 1: 2 MFMA + 16 v_max3_i32 + cmp + branch", d_ab, d_sink, d_clk);
    run_cost<2>("mode 2: 2 MFMA + cvt_2xpk16_fp6 + 7 and/and_or + cmp + branch", d_ab, d_sink, d_clk);
    run_cost<3>("mode 3: 2 MFMA + cvt_2xpk16_fp6 alone", d_ab, d_sink, d_clk);
    run_cost<4>("mode 4: 2 MFMA + chain of 16 v_bitop3_b32 (AND) + cmp + branch", d_ab, d_sink, d_clk);
    run_cost<5>("mode 5: 2 MFMA + tree of 16 v_bitop3_b32 (AND) + cmp + branch", d_ab, d_sink, d_clk);
    run_cost<6>("mode 6: mode 4 with the A operand read from LDS each trip (read, wait, multiply)", d_ab, d_sink, d_clk);
    run_cost<7>("mode 7: mode 4 with the NEXT trip's A operand read before this trip's inspection", d_ab, d_sink, d_clk);
    return 0;
}
