// valu_rate.hip -- cycles one SIMD spends per wave64 VALU instruction, by opcode, with 1 / 2 / 4 waves per SIMD (gfx950).
// Each wave runs `trips` passes over 16 independent instructions of one opcode; wall time x 2.4 GHz / instructions per SIMD.
// The clock is not pinned: "MFMA-free integer VALU" runs near 2.4 GHz; compare the columns, not the third digit.
// Build: hipcc -O2 --offload-arch=gfx950 valu_rate.hip -o valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

#define OP3(name, text)                                                                                             \
    template <> struct Op<name> {                                                                                     \
        static __device__ __forceinline__ void run(unsigned int &d, unsigned int a, unsigned int b) {              \
            asm volatile(text : "+v"(d) : "v"(a), "v"(b));                                                          \
        }                                                                                                             \
    };

template <int N> struct Op;
OP3(0, "v_add_f32 %0, %1, %2")
OP3(1, "v_max_f32 %0, %1, %2")
OP3(2, "v_and_b32 %0, %1, %2")
OP3(3, "v_max3_f32 %0, %0, %1, %2")
OP3(4, "v_or3_b32 %0, %0, %1, %2")
OP3(5, "v_and_or_b32 %0, %0, %1, %2")
OP3(6, "v_perm_b32 %0, %0, %1, %2")
OP3(7, "v_bfe_u32 %0, %1, 3, 8")
OP3(8, "v_lshl_add_u32 %0, %1, 2, %2")
OP3(9, "v_fma_f32 %0, %1, %2, %0")
OP3(10, "v_pk_max_i16 %0, %1, %2")
OP3(11, "v_lshlrev_b32 %0, 3, %1")
OP3(12, "v_add_u32 %0, %1, %2")
OP3(13, "v_min3_i32 %0, %0, %1, %2")
OP3(14, "v_max_i32 %0, %1, %2")
OP3(15, "v_mov_b32 %0, %1")
OP3(16, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80")
OP3(17, "v_or_b32 %0, %1, %2")
OP3(18, "v_mul_f32 %0, %1, %2")
OP3(19, "v_add3_u32 %0, %0, %1, %2")
OP3(20, "v_mad_u32_u24 %0, %1, %2, %0")
OP3(21, "v_xad_u32 %0, %0, %1, %2")
OP3(22, "v_alignbit_b32 %0, %0, %1, 31")
OP3(23, "v_bfi_b32 %0, %1, %2, %0")
OP3(24, "v_cmp_lt_f32 vcc, %1, %2")
OP3(25, "v_fma_f32 %0, %1, %2, %0 clamp")
OP3(26, "v_add_f32 %0, |%1|, -%2")
OP3(27, "v_min_f32 %0, %1, %2")
OP3(28, "v_maximum3_f32 %0, %0, %1, %2")
OP3(29, "v_med3_f32 %0, %0, %1, %2")
OP3(30, "v_cndmask_b32 %0, %1, %2, vcc")
OP3(31, "v_lshl_or_b32 %0, %0, 1, %1")

template <int N>
__global__ void __launch_bounds__(1024) valu_kernel(unsigned int *out, int trips, unsigned long long *cyc) {
    unsigned int v[16];
    __syncthreads();
    const long long t0 = clock64();
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 16 + i;
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int k = 0; k < 16; k++) Op<N>::run(v[k], v[(k + 5) % 16], v[(k + 9) % 16]);
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) atomicMax(cyc, (unsigned long long) (t1 - t0));       // the slowest wave of the launch
    unsigned int s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s ^= v[i];
    if (s == 0x12345678u) out[0] = s;
}

#define OP64(name, text)                                                                                            \
    template <> struct Op64<name> {                                                                                   \
        static __device__ __forceinline__ void run(unsigned long long &d, unsigned long long a, unsigned long long b) { \
            asm volatile(text : "+v"(d) : "v"(a), "v"(b));                                                          \
        }                                                                                                             \
    };
template <int N> struct Op64;
OP64(0, "v_pk_add_f32 %0, %1, %2")
OP64(1, "v_pk_fma_f32 %0, %1, %2, %0")
OP64(2, "v_pk_mul_f32 %0, %1, %2")
OP64(3, "v_max_f64 %0, %1, %2")
OP64(4, "v_add_f64 %0, %1, %2")
OP64(5, "v_pk_mov_b32 %0, %1, %2")
OP64(6, "v_fma_f64 %0, %1, %2, %0")
OP64(7, "v_lshlrev_b64 %0, 3, %1")

template <int N>
__global__ void __launch_bounds__(1024) valu64_kernel(unsigned int *out, int trips, unsigned long long *cyc) {
    unsigned long long v[16];
    __syncthreads();
    const long long t0 = clock64();
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = 0x3F8000003F800000ULL + threadIdx.x * 16 + i;
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int k = 0; k < 16; k++) Op64<N>::run(v[k], v[(k + 5) % 16], v[(k + 9) % 16]);
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) atomicMax(cyc, (unsigned long long) (t1 - t0));
    unsigned long long s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s ^= v[i];
    if (s == 0x12345678u) out[0] = (unsigned int) s;
}

template <int N, bool W64 = false>
static void run(const char *what, unsigned int *d_out) {
    const int trips = 100000;
    static unsigned long long *d_cyc = nullptr;
    if (!d_cyc) CK(hipMalloc(&d_cyc, 8));
    printf("%-16s", what);
    for (int threads : {256, 512, 1024}) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto launch = [&]() {
            if constexpr (W64) hipLaunchKernelGGL((valu64_kernel<N>), dim3(256), dim3(threads), 0, 0, d_out, trips, d_cyc);
            else hipLaunchKernelGGL((valu_kernel<N>), dim3(256), dim3(threads), 0, 0, d_out, trips, d_cyc);
        };
        launch();
        CK(hipMemset(d_cyc, 0, 8));
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long cyc = 0;
        CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        const int wps = threads / 256;
        printf("  %dw/SIMD %5.2f cyc/instr (%4.0f MHz)", wps, (double) cyc / ((double) trips * 16 * wps), cyc / (ms * 1e3));
    }
    printf("\n");
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned int *d_out;
    CK(hipMalloc(&d_out, 64));
    run<0>("v_add_f32", d_out);
    run<1>("v_max_f32", d_out);
    run<9>("v_fma_f32", d_out);
    run<3>("v_max3_f32", d_out);
    run<2>("v_and_b32", d_out);
    run<12>("v_add_u32", d_out);
    run<14>("v_max_i32", d_out);
    run<11>("v_lshlrev_b32", d_out);
    run<15>("v_mov_b32", d_out);
    run<4>("v_or3_b32", d_out);
    run<5>("v_and_or_b32", d_out);
    run<13>("v_min3_i32", d_out);
    run<6>("v_perm_b32", d_out);
    run<7>("v_bfe_u32", d_out);
    run<8>("v_lshl_add_u32", d_out);
    run<10>("v_pk_max_i16", d_out);
    run<16>("v_bitop3 (and)", d_out);
    run<17>("v_or_b32", d_out);
    run<18>("v_mul_f32", d_out);
    run<19>("v_add3_u32", d_out);
    run<20>("v_mad_u32_u24", d_out);
    run<21>("v_xad_u32", d_out);
    run<22>("v_alignbit_b32", d_out);
    run<23>("v_bfi_b32", d_out);
    run<24>("v_cmp_lt_f32", d_out);
    run<25>("v_fma_f32 clamp", d_out);
    run<26>("v_add_f32 |a|,-b", d_out);
    run<27>("v_min_f32", d_out);
    run<28>("v_maximum3_f32", d_out);
    run<29>("v_med3_f32", d_out);
    run<30>("v_cndmask_b32", d_out);
    run<31>("v_lshl_or_b32", d_out);
    run<0, true>("v_pk_add_f32", d_out);
    run<1, true>("v_pk_fma_f32", d_out);
    run<2, true>("v_pk_mul_f32", d_out);
    run<3, true>("v_max_f64", d_out);
    run<4, true>("v_add_f64", d_out);
    run<5, true>("v_pk_mov_b32", d_out);
    run<6, true>("v_fma_f64", d_out);
    run<7, true>("v_lshlrev_b64", d_out);
    return 0;
}
