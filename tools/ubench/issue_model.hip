// issue_model.hip -- what one SIMD sustains for the pre-filter's instruction mix (gfx950).
// Per loop trip a wave issues 1 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, fp6 x fp4: 32 cycles of the matrix pipe) and K plain VALU
// instructions (v_max3_f32 on independent registers), optionally S SALU instructions and L ds_read_b64.  Blocks of 256 / 512 / 1024
// threads put 1 / 2 / 4 waves on each SIMD.  Reported: cycles per trip PER SIMD = wall time x shader clock / (trips x waves per SIMD),
// so that "32" = matrix pipe saturated, and K x 2 or K x 4 tells what a VALU instruction costs beside it.  The shader clock is
// s_memtime against the 100 MHz s_memrealtime inside the kernel.  (Round 2 divided ONE wave's own cycle count by the waves per SIMD:
// the oldest wave of a SIMD wins the arbitration and finishes early, so that column fell below the matrix pipe's 32-cycle floor.)
// Build: hipcc -O2 --offload-arch=gfx950 issue_model.hip -o issue_model.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int K, int S, int L, bool MFMA>
__global__ void __launch_bounds__(1024) mix_kernel(float *out, int trips, long long *cycles) {
    __shared__ unsigned long long lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    v8i a = {1, 2, 3, 4, 5, 6, 0, 0}, b = {1, 2, 3, 4, 0, 0, 0, 0};
    v16f acc0 = {}, acc1 = {};
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = (float) (threadIdx.x * 16 + i);
    unsigned int s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    unsigned long long l0 = 0, l1 = 0;
    const unsigned int laddr = (unsigned int) (size_t) (lds + (threadIdx.x & 63));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < trips; t += 2) {
#pragma unroll
        for (int h = 0; h < 2; h++) {                   // two trips per pass, one accumulator each: no copies, no branch on t
            if (MFMA) {
                if (h) acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc1, 2, 4, 0, 127, 0, 127);
                else acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc0, 2, 4, 0, 127, 0, 127);
            }
#pragma unroll
            for (int k = 0; k < K; k++)
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[k % 16]) : "v"(v[(k + 5) % 16]), "v"(v[(k + 9) % 16]));
#pragma unroll
            for (int k = 0; k < S; k++) {
                if ((k & 3) == 0) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
                if ((k & 3) == 1) asm volatile("s_xor_b32 %0, %0, %1" : "+s"(s1) : "s"(s2) : "scc");
                if ((k & 3) == 2) asm volatile("s_lshl_b32 %0, %1, 1" : "+s"(s2) : "s"(s3) : "scc");
                if ((k & 3) == 3) asm volatile("s_and_b32 %0, %0, %1" : "+s"(s3) : "s"(s0) : "scc");
            }
#pragma unroll
            for (int k = 0; k < L; k++) {
                if (k & 1) asm volatile("ds_read_b64 %0, %1" : "=v"(l1) : "v"(laddr));
                else asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(l0) : "v"(laddr));
            }
            if (L) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += v[i] + acc0[i] + acc1[i];
    s += (float) (s0 + s1 + s2 + s3) + (float) (l0 + l1);
    if (s == 1234.5f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cycles[0] = (long long) (t1 - t0); cycles[1] = (long long) (r1 - r0); }
}

template <int K, int S, int L, bool MFMA>
static void run(const char *what, float *d_out, long long *d_cyc) {
    const int trips = 20000;
    printf("%-44s", what);
    for (int threads : {256, 512, 1024}) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL((mix_kernel<K, S, L, MFMA>), dim3(256), dim3(threads), 0, 0, d_out, trips, d_cyc);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((mix_kernel<K, S, L, MFMA>), dim3(256), dim3(threads), 0, 0, d_out, trips, d_cyc);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        long long cyc[2] = {0, 0};
        CK(hipMemcpy(cyc, d_cyc, 16, hipMemcpyDeviceToHost));
        const int wps = threads / 256;
        const double mhz = cyc[1] ? 100.0 * (double) cyc[0] / (double) cyc[1] : 0.0;
        printf("  %dw/SIMD: %5.1f cycles per SIMD-trip (%.3f ms at %4.0f MHz)", wps, ms * 1e-3 * mhz * 1e6 / ((double) trips * wps), ms, mhz);
    }
    printf("\n");
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *d_out;
    long long *d_cyc;
    CK(hipMalloc(&d_out, 64));
    CK(hipMalloc(&d_cyc, 64));
    run<0, 0, 0, true>("MFMA only", d_out, d_cyc);
    run<8, 0, 0, false>("8 VALU only", d_out, d_cyc);
    run<16, 0, 0, false>("16 VALU only", d_out, d_cyc);
    run<4, 0, 0, true>("MFMA + 4 VALU", d_out, d_cyc);
    run<8, 0, 0, true>("MFMA + 8 VALU", d_out, d_cyc);
    run<12, 0, 0, true>("MFMA + 12 VALU", d_out, d_cyc);
    run<16, 0, 0, true>("MFMA + 16 VALU", d_out, d_cyc);
    run<12, 8, 0, true>("MFMA + 12 VALU + 8 SALU", d_out, d_cyc);
    run<12, 8, 5, true>("MFMA + 12 VALU + 8 SALU + 5 ds_read_b64", d_out, d_cyc);
    run<12, 0, 5, true>("MFMA + 12 VALU + 5 ds_read_b64", d_out, d_cyc);
    run<6, 4, 5, true>("MFMA + 6 VALU + 4 SALU + 5 ds_read_b64", d_out, d_cyc);
    return 0;
}
