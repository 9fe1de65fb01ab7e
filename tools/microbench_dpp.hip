// microbench_dpp.hip -- cost and semantics of a full-wave rotate by one lane on gfx950:
// v_mov_b32_dpp wave_ror:1 / wave_rol:1 / row_ror:1, against plain v_mov_b32.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITERS; ++i) {
        if (KIND == 0)
            asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if (KIND == 1)
            asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %4 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if (KIND == 2)
            asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if (KIND == 3) {  // rotate fused into an add: v_add_f32_dpp
            asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %3 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %4, %5 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %6, %7 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %1, %1, %0 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %2 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %4 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %6 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ void semantics(int* out) {  // one wave: where does lane l's value go?
    int v = threadIdx.x;
    int r, l, rr;
    asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 wave_rol:1 row_mask:0xf bank_mask:0xf" : "=v"(l) : "v"(v));
    asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(rr) : "v"(v));
    out[threadIdx.x] = r; out[64 + threadIdx.x] = l; out[128 + threadIdx.x] = rr;
}

template <int KIND>
int run(const char* name, int w, float* out, unsigned long long* clk) {
    int blocks = 256 * w;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    double ghz = double(h[0]) / double(h[1]) * 0.1, us = ms * 1e3 / 5;
    printf("%-24s waves/SIMD=%d %8.1f us clock %.2f GHz  %.2f cycles per wave-instruction per SIMD\n", name, w, us, ghz, us * 1e3 * ghz / (double(ITERS) * 8 * w));
    return 0;
}

int main() {
    float* out; unsigned long long* clk; int* sem;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * 8 * 256)); CHECK(hipMalloc(&clk, 16)); CHECK(hipMalloc(&sem, 192 * 4));
    hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, sem);
    int h[192]; CHECK(hipMemcpy(h, sem, sizeof(h), hipMemcpyDeviceToHost));
    printf("wave_ror:1  lane0<-%d lane1<-%d lane15<-%d lane16<-%d lane31<-%d lane32<-%d lane63<-%d\n", h[0], h[1], h[15], h[16], h[31], h[32], h[63]);
    printf("wave_rol:1  lane0<-%d lane1<-%d lane15<-%d lane16<-%d lane31<-%d lane32<-%d lane63<-%d\n", h[64], h[65], h[79], h[80], h[95], h[96], h[127]);
    printf("row_ror:1   lane0<-%d lane1<-%d lane15<-%d lane16<-%d lane31<-%d lane32<-%d lane63<-%d\n", h[128], h[129], h[143], h[144], h[159], h[160], h[191]);
    for (int w : {2, 4, 8}) {
        if (run<0>("v_mov_b32", w, out, clk)) return 1;
        if (run<1>("v_mov_b32_dpp wave_ror:1", w, out, clk)) return 1;
        if (run<2>("v_mov_b32_dpp row_ror:1", w, out, clk)) return 1;
        if (run<3>("v_add_f32_dpp wave_ror:1", w, out, clk)) return 1;
    }
    return 0;
}
