// microbench_valu.hip -- issue rates of the fp32 VALU instructions the all-pairs kernel is made
// of, on the whole chip.  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o ...
// Prints wave-instructions per SIMD per microsecond and the implied cycles per wave-instruction
// at the clock measured with s_memtime/s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float b, float c, unsigned long long* clk) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float float2_ __attribute__((ext_vector_type(2)));
    float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    float2_ bb = {b, b}, cc = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITERS; ++i) {
        if (KIND == 0) {  // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 1) {  // 8 independent v_pk_fma_f32 (16 fma)
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(bb), "v"(cc));
        } else if (KIND == 2) {  // 8 independent v_rsq_f32
            asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                         "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 3) {  // the pair body unpacked: 3 sub, 3 fma, 1 rsq, 3 mul, 3 fma  (x2 pairs)
            asm volatile(
                "v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n"
                "v_fma_f32 %3, %0, %0, %9\n v_fma_f32 %3, %1, %1, %3\n v_fma_f32 %3, %2, %2, %3\n"
                "v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n"
                "v_fma_f32 %7, %4, %4, %9\n v_fma_f32 %7, %5, %5, %7\n v_fma_f32 %7, %6, %6, %7\n"
                "v_rsq_f32 %3, %3\n v_rsq_f32 %7, %7\n"
                "v_mul_f32 %3, %3, %3\n v_mul_f32 %3, %3, %9\n v_mul_f32 %3, %3, %3\n"
                "v_mul_f32 %7, %7, %7\n v_mul_f32 %7, %7, %9\n v_mul_f32 %7, %7, %7\n"
                "v_fma_f32 %0, %0, %3, %0\n v_fma_f32 %1, %1, %3, %1\n v_fma_f32 %2, %2, %3, %2\n"
                "v_fma_f32 %4, %4, %7, %4\n v_fma_f32 %5, %5, %7, %5\n v_fma_f32 %6, %6, %7, %6\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 4) {  // the pair body packed over 2 bodies: 3 pk_add, 3 pk_fma, 2 rsq, 3 pk_mul, 3 pk_fma
            asm volatile(
                "v_pk_add_f32 %0, %8, %0\n v_pk_add_f32 %1, %8, %1\n v_pk_add_f32 %2, %8, %2\n"
                "v_pk_fma_f32 %3, %0, %0, %9\n v_pk_fma_f32 %3, %1, %1, %3\n v_pk_fma_f32 %3, %2, %2, %3\n"
                "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n"
                "v_pk_mul_f32 %3, %3, %3\n v_pk_mul_f32 %3, %3, %9\n v_pk_mul_f32 %3, %3, %3\n"
                "v_pk_fma_f32 %0, %0, %3, %0\n v_pk_fma_f32 %1, %1, %3, %1\n v_pk_fma_f32 %2, %2, %3, %2\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(a4), "+v"(a5), "+v"(p6), "+v"(p7) : "v"(bb), "v"(cc));
        } else if (KIND == 6) {  // 8 independent v_fmac_f32 in the 4-byte VOP2 encoding
            asm volatile("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
                         "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 7) {  // the same in the 8-byte VOP3 encoding
            asm volatile("v_fmac_f32_e64 %0, %8, %9\n v_fmac_f32_e64 %1, %8, %9\n v_fmac_f32_e64 %2, %8, %9\n v_fmac_f32_e64 %3, %8, %9\n"
                         "v_fmac_f32_e64 %4, %8, %9\n v_fmac_f32_e64 %5, %8, %9\n v_fmac_f32_e64 %6, %8, %9\n v_fmac_f32_e64 %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 8) {  // 8 independent v_mul_f32 (VOP2)
            asm volatile("v_mul_f32_e32 %0, %8, %0\n v_mul_f32_e32 %1, %8, %1\n v_mul_f32_e32 %2, %8, %2\n v_mul_f32_e32 %3, %8, %3\n"
                         "v_mul_f32_e32 %4, %8, %4\n v_mul_f32_e32 %5, %8, %5\n v_mul_f32_e32 %6, %8, %6\n v_mul_f32_e32 %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 9) {  // v_fma_f32 with the accumulator as src2 (what fma(-x, s, acc) compiles to)
            asm volatile("v_fma_f32 %0, -%8, %9, %0\n v_fma_f32 %1, -%8, %9, %1\n v_fma_f32 %2, -%8, %9, %2\n v_fma_f32 %3, -%8, %9, %3\n"
                         "v_fma_f32 %4, -%8, %9, %4\n v_fma_f32 %5, -%8, %9, %5\n v_fma_f32 %6, -%8, %9, %6\n v_fma_f32 %7, -%8, %9, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 10) {  // v_fma_f32 d, x, x, c (the r2 start)
            asm volatile("v_fma_f32 %0, %1, %1, %9\n v_fma_f32 %1, %2, %2, %9\n v_fma_f32 %2, %3, %3, %9\n v_fma_f32 %3, %4, %4, %9\n"
                         "v_fma_f32 %4, %5, %5, %9\n v_fma_f32 %5, %6, %6, %9\n v_fma_f32 %6, %7, %7, %9\n v_fma_f32 %7, %0, %0, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 11) {  // v_fma_f32 with three different varying sources
            asm volatile("v_fma_f32 %0, %1, %2, %3\n v_fma_f32 %1, %2, %3, %4\n v_fma_f32 %2, %3, %4, %5\n v_fma_f32 %3, %4, %5, %6\n"
                         "v_fma_f32 %4, %5, %6, %7\n v_fma_f32 %5, %6, %7, %0\n v_fma_f32 %6, %7, %0, %1\n v_fma_f32 %7, %0, %1, %2\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 5) {  // 4 fma + 1 rsq interleaved (does the transcendental co-issue?)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_rsq_f32 %4, %4\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_rsq_f32 %7, %7\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.y + p7.x;
}

template <int KIND>
int run(const char* name, int instr_per_iter, int waves_per_simd, float* out, unsigned long long* clk) {
    int blocks = 256 * waves_per_simd;  // 256 CUs x 4 SIMDs x waves_per_simd waves / 4 waves per block
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    double ghz = double(h[0]) / double(h[1]) * 0.1;  // s_memrealtime ticks at 100 MHz
    double us = ms * 1e3 / reps;
    double winstr_per_simd = double(ITERS) * instr_per_iter * waves_per_simd;
    double cyc = us * 1e3 * ghz / winstr_per_simd;
    printf("%-28s waves/SIMD=%d  %8.1f us  clock %.2f GHz  %.2f cycles per wave-instruction per SIMD\n", name, waves_per_simd, us, ghz, cyc);
    return 0;
}

int main() {
    float* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * 8 * 256));
    CHECK(hipMalloc(&clk, 16));
    for (int w : {3, 8}) {
        if (run<0>("v_fma_f32", 8, w, out, clk)) return 1;
        if (run<6>("v_fmac_f32 VOP2 (4 B)", 8, w, out, clk)) return 1;
        if (run<7>("v_fmac_f32 VOP3 (8 B)", 8, w, out, clk)) return 1;
        if (run<8>("v_mul_f32 VOP2 (4 B)", 8, w, out, clk)) return 1;
        if (run<9>("v_fma_f32 d,-b,c,d", 8, w, out, clk)) return 1;
        if (run<10>("v_fma_f32 d,x,x,c", 8, w, out, clk)) return 1;
        if (run<11>("v_fma_f32 d,x,y,z", 8, w, out, clk)) return 1;
        if (run<1>("v_pk_fma_f32", 8, w, out, clk)) return 1;
        if (run<2>("v_rsq_f32", 8, w, out, clk)) return 1;
        if (run<5>("4 fma : 1 rsq mix (10 instr)", 10, w, out, clk)) return 1;
        if (run<3>("pair body x2 unpacked (26)", 26, w, out, clk)) return 1;
        if (run<4>("pair body x2 packed (14)", 14, w, out, clk)) return 1;
    }
    return 0;
}
