// Probe (measurement only): v_mfma_scale_f32_32x32x64_f8f6f4 with A in fp6 (e2m3) and B in fp4 (e2m1) on gfx950:
//   (1) operand layout, checked with exact small-integer data against a host product,
//   (2) cycles per instruction on one SIMD against v_mfma_i32_32x32x32_i8 (K = 32), same loop shape.
// Hypothesis under test: lane l holds row / column l & 31 and the 32 consecutive k = 32 (l >> 5) + j, j = 0..31, packed
// little-endian (fp6: value j at bits [6j, 6j+6) of the lane's 192 bits; fp4: bits [4j, 4j+4) of 128 bits); scale 127 = 2^0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const i32x8 *a, const i32x8 *b, float *d, int fa, int fb) {
    const int l = threadIdx.x;
    f32x16 c = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    if (fa == 2 && fb == 4) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, 2, 4, 0, 127, 0, 127);
    else if (fa == 2 && fb == 2) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, 2, 2, 0, 127, 0, 127);
    else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, 4, 4, 0, 127, 0, 127);
    for (int j = 0; j < 16; j++) d[l * 16 + j] = c[j];
}

template <int MODE>
__global__ void __launch_bounds__(1024) rate(int iters, float *out, unsigned long long *cyc) {
    i32x8 a = {(int) threadIdx.x * 0x01010101, 0x11111111, 0x22222222, 0x12345678, 0x0f0f0f0f, 0x33333333, 0, 0};
    i32x8 b = {0x22222222, 0x02020202, 0x20202020, (int) threadIdx.x, 0, 0, 0, 0};
    i32x4 ai = {(int) threadIdx.x, 1, 2, 3}, bi = {1, 0x100, 0x10000, 1};
    f32x16 c0 = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
    i32x16 z = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}, d0 = z, d1 = z, d2 = z, d3 = z;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 2, 4, 0, 127, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 2, 4, 0, 127, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 2, 4, 0, 127, 0, 127);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 2, 4, 0, 127, 0, 127);
        } else if (MODE == 1) {
            d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, bi, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, bi, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, bi, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, bi, d3, 0, 0, 0);
        } else if (MODE == 2) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 2, 2, 0, 127, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 2, 2, 0, 127, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 2, 2, 0, 127, 0, 127);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 2, 2, 0, 127, 0, 127);
        } else {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 127, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 127, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 4, 4, 0, 127, 0, 127);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 4, 4, 0, 127, 0, 127);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int j = 0; j < 16; j++) r += c0[j] + c1[j] + c2[j] + c3[j] + (float) (d0[j] + d1[j] + d2[j] + d3[j]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) atomicMax(&cyc[blockIdx.x], t1 - t0);
}

static float fp6_val(int code) {                 // e2m3: s eemmm, bias 1
    const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
    const float v = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, e - 1);
    return s ? -v : v;
}
static float fp4_val(int code) {                 // e2m1: s eem, bias 1
    const int s = (code >> 3) & 1, e = (code >> 1) & 3, m = code & 1;
    const float v = e == 0 ? m * 0.5f : ldexpf(1.0f + m * 0.5f, e - 1);
    return s ? -v : v;
}
static void put_bits(unsigned *w, int bit, int nbits, unsigned v) {
    for (int i = 0; i < nbits; i++) if ((v >> i) & 1) w[(bit + i) >> 5] |= 1u << ((bit + i) & 31);
}

int main() {
    // ---- layout ----
    for (int pass = 0; pass < 3; pass++) {
        const int fa = pass == 2 ? 4 : 2, fb = pass == 1 ? 2 : 4;
        const int abits = fa == 2 ? 6 : 4, bbits = fb == 2 ? 6 : 4;
        static int Acode[32][64], Bcode[64][32];
        srand(7 + pass);
        for (int r = 0; r < 32; r++) for (int k = 0; k < 64; k++) Acode[r][k] = rand() % (1 << abits);
        for (int k = 0; k < 64; k++) for (int c = 0; c < 32; c++) Bcode[k][c] = rand() % (1 << bbits);
        unsigned ha[64][8], hb[64][8];
        memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
        for (int l = 0; l < 64; l++)
            for (int j = 0; j < 32; j++) {
                put_bits(ha[l], abits * j, abits, (unsigned) Acode[l & 31][32 * (l >> 5) + j]);
                put_bits(hb[l], bbits * j, bbits, (unsigned) Bcode[32 * (l >> 5) + j][l & 31]);
            }
        i32x8 *da, *db; float *dd;
        hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, 64 * 16 * 4);
        hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd, fa, fb);
        float hd[64 * 16];
        hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++)
            for (int j = 0; j < 16; j++) {
                const int row = (j & 3) + 8 * (j >> 2) + 4 * (l >> 5), col = l & 31;
                double want = 0;
                for (int k = 0; k < 64; k++) want += (double) (fa == 2 ? fp6_val(Acode[row][k]) : fp4_val(Acode[row][k])) *
                                                     (double) (fb == 2 ? fp6_val(Bcode[k][col]) : fp4_val(Bcode[k][col]));
                if (fabs(want - hd[l * 16 + j]) > 1e-3) { if (bad < 5) printf("  mismatch lane %d reg %d: got %g want %g\n", l, j, hd[l * 16 + j], want); bad++; }
            }
        printf("layout A=%s B=%s: lane l = row/col l&31, k = 32*(l>>5)+j, little-endian packing, scale 127: %s (%d of 1024 wrong)\n",
               fa == 2 ? "fp6(e2m3)" : "fp4(e2m1)", fb == 2 ? "fp6(e2m3)" : "fp4(e2m1)", bad ? "WRONG" : "CONFIRMED", bad);
    }
    // ---- rate ----
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 20000;
    const char *names[4] = {"fp6 x fp4  32x32x64 (scaled f8f6f4)", "int8       32x32x32", "fp6 x fp6  32x32x64", "fp4 x fp4  32x32x64"};
    for (int threads : {256, 1024}) for (int mode = 0; mode < 4; mode++) {
        hipMemset(cyc, 0, 256 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(threads), 0, 0, iters, out, cyc);
            if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(threads), 0, 0, iters, out, cyc);
            if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(256), dim3(threads), 0, 0, iters, out, cyc);
            if (mode == 3) hipLaunchKernelGGL(rate<3>, dim3(256), dim3(threads), 0, 0, iters, out, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        unsigned long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
        const double per = (double) mx / (4.0 * iters) / (threads / 256);
        printf("%-40s threads %4d: %.1f cycles per MFMA per SIMD (slowest wave), wall %.3f ms = %.2f ns per MFMA per SIMD -> %.0f TOP/s chip at K=%d\n",
               names[mode], threads, per, ms, ms * 1e6 / (4.0 * iters * (threads / 256)), 1024.0 * 2 * 32 * 32 * (mode == 1 ? 32 : 64) / (ms * 1e6 / (4.0 * iters * (threads / 256))) * 1e-3, mode == 1 ? 32 : 64);
    }
    return 0;
}
