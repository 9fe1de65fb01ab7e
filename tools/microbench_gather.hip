// microbench_gather.hip -- the ceiling of the Barnes-Hut walk's memory pattern on MI355X.
//
// k_bh_walk does, per lane and per visit, two 16-byte loads of one 32-byte node record at an address that
// depends on the previous record (pointer chase), different lanes on different records.  This measures
// how many such visits per cycle per CU the chip sustains
//   (a) from global memory (L1/L2-resident footprint, like the 3 MiB node array), lanes fully divergent,
//       in groups of g lanes on the same record, or all on the same record (coalesced);
//   (b) from an LDS table (ds_read_b128 x 2 at per-lane addresses), same coherence classes;
//   (c) mixed: a share of the steps from LDS, the rest from global (the staged-top-of-tree walk).
// Output: one line per case: visits / cycle / CU at the clock the run held (s_memtime vs s_memrealtime).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_gather.hip -o build/microbench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

struct alignas(32) Rec { float4 a; float4 b; };  // b.y = next index (int bits)

// global chase: `iters` dependent visits per lane
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_chase_global(const Rec* __restrict__ t, const int* __restrict__ start, int iters,
                                                        float* __restrict__ out, unsigned long long* __restrict__ clk) {
    const int gid = blockIdx.x * BLOCK + threadIdx.x;
    int i = start[gid];
    float s = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < iters; ++k) {
        const float4 A = t[i].a;
        const float4 B = t[i].b;
        asm volatile("" :: "v"(A.w), "v"(B.x));
        s += A.x;
        i = __float_as_int(B.y);
    }
    if (gid == 0) { clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    out[gid] = s + float(i);
}

// LDS chase: the first M records staged in LDS (16 B per lane per copy step), chain confined to [0, M)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_chase_lds(const Rec* __restrict__ t, int M, const int* __restrict__ start, int iters,
                                                     float* __restrict__ out, unsigned long long* __restrict__ clk) {
    extern __shared__ float4 lds[];
    for (int k = threadIdx.x; k < 2 * M; k += BLOCK) lds[k] = reinterpret_cast<const float4*>(t)[k];
    __syncthreads();
    const int gid = blockIdx.x * BLOCK + threadIdx.x;
    int i = start[gid];
    float s = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < iters; ++k) {
        const float4 A = lds[2 * i];
        const float4 B = lds[2 * i + 1];
        asm volatile("" :: "v"(A.w), "v"(B.x));
        s += A.x;
        i = __float_as_int(B.y);
    }
    if (gid == 0) { clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    out[gid] = s + float(i);
}

// mixed chase: index < M -> LDS copy of the record, else global (the record's link decides where the next
// visit goes: the table is built so that a given share of the links point below M)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_chase_mixed(const Rec* __restrict__ t, int M, const int* __restrict__ start, int iters,
                                                       float* __restrict__ out, unsigned long long* __restrict__ clk) {
    extern __shared__ float4 lds[];
    for (int k = threadIdx.x; k < 2 * M; k += BLOCK) lds[k] = reinterpret_cast<const float4*>(t)[k];
    __syncthreads();
    const int gid = blockIdx.x * BLOCK + threadIdx.x;
    int i = start[gid];
    float s = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < iters; ++k) {
        float4 A, B;
        if (i < M) { A = lds[2 * i]; B = lds[2 * i + 1]; }
        else { A = t[i].a; B = t[i].b; }
        asm volatile("" :: "v"(A.w), "v"(B.x));
        s += A.x;
        i = __float_as_int(B.y);
    }
    if (gid == 0) { clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    out[gid] = s + float(i);
}

int main(int argc, char** argv) {
    const int n_rec = argc > 1 ? atoi(argv[1]) : 98304;      // 3 MiB of records, like N = 65 536 bodies
    const int waves = 8192, iters = 2000;
    const int lanes = waves * 64;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::mt19937 rng(12345);
    Rec* d_t; int* d_start; float* d_out; unsigned long long* d_clk;
    CHECK(hipMalloc(&d_t, size_t(n_rec) * sizeof(Rec)));
    CHECK(hipMalloc(&d_start, lanes * sizeof(int)));
    CHECK(hipMalloc(&d_out, lanes * sizeof(float)));
    CHECK(hipMalloc(&d_clk, 2 * sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<Rec> tab(n_rec);
    std::vector<int> start(lanes);

    // links: record i -> uniformly random record in [lo(i), hi(i))
    auto fill = [&](int M, double share_lds) {  // share_lds of ALL links point below M (M = 0: none)
        std::uniform_real_distribution<double> u(0.0, 1.0);
        for (int i = 0; i < n_rec; ++i) {
            int nxt;
            if (M > 0 && u(rng) < share_lds) nxt = int(u(rng) * M);
            else nxt = (M > 0 && share_lds >= 1.0) ? int(u(rng) * M) : M + int(u(rng) * (n_rec - M));
            if (nxt >= n_rec) nxt = n_rec - 1;
            tab[i].a = make_float4(float(i), 1.f, 2.f, 3.f);
            tab[i].b = make_float4(0.5f, __builtin_bit_cast(float, nxt), 0.f, 0.f);
        }
    };
    auto starts = [&](int group, int range) {  // lanes in groups of `group` share a start (and so a whole chain)
        std::uniform_int_distribution<int> d(0, range - 1);
        for (int l = 0; l < lanes; l += group) { const int s = d(rng); for (int q = 0; q < group && l + q < lanes; ++q) start[l + q] = s; }
    };
    auto report = [&](const char* what, float ms) {
        unsigned long long clk[2];
        (void)hipMemcpy(clk, d_clk, sizeof(clk), hipMemcpyDeviceToHost);
        const double ghz = clk[1] ? double(clk[0]) / (double(clk[1]) * 10.0) : 0.0;  // s_memrealtime ticks at 100 MHz
        const double visits = double(lanes) * iters;
        const double cyc = ms * 1e-3 * ghz * 1e9;
        printf("%-58s %7.3f ms  %5.2f GHz  %6.3f visits/cycle/CU  (%5.1f cycles per wave-visit per CU)\n", what, ms, ghz,
               visits / cyc / cus, cyc * cus / (double(waves) * iters));
    };
    printf("%d CUs, %d records (%.1f MiB), %d waves x %d dependent visits, 2 x 16 B per visit\n", cus, n_rec,
           n_rec * 32.0 / 1048576.0, waves, iters);
    float ms;
    // (a) global
    fill(0, 0.0);
    CHECK(hipMemcpy(d_t, tab.data(), size_t(n_rec) * sizeof(Rec), hipMemcpyHostToDevice));
    const int groups[] = {1, 2, 4, 8, 16, 64};
    for (int blk : {64, 256}) {
        for (int g : groups) {
            starts(g, n_rec);
            CHECK(hipMemcpy(d_start, start.data(), lanes * sizeof(int), hipMemcpyHostToDevice));
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                if (blk == 64) hipLaunchKernelGGL(k_chase_global<64>, dim3(lanes / 64), dim3(64), 0, 0, d_t, d_start, iters, d_out, d_clk);
                else hipLaunchKernelGGL(k_chase_global<256>, dim3(lanes / 256), dim3(256), 0, 0, d_t, d_start, iters, d_out, d_clk);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
            }
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            char what[128];
            snprintf(what, sizeof what, "global, %d-thread workgroups, %2d lanes per record", blk, g);
            report(what, ms);
        }
    }
    // (b) LDS and (c) mixed, 1024-thread workgroups (16 waves share one table)
    for (int M : {1024, 2048, 4096}) {
        const size_t lds_bytes = size_t(M) * 32;
        if (lds_bytes > 64 * 1024) {
            CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chase_lds<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
            CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chase_mixed<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
        }
        fill(M, 1.0);
        CHECK(hipMemcpy(d_t, tab.data(), size_t(n_rec) * sizeof(Rec), hipMemcpyHostToDevice));
        for (int g : {1, 4, 64}) {
            starts(g, M);
            CHECK(hipMemcpy(d_start, start.data(), lanes * sizeof(int), hipMemcpyHostToDevice));
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_chase_lds<1024>, dim3(lanes / 1024), dim3(1024), lds_bytes, 0, d_t, M, d_start, iters, d_out, d_clk);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
            }
            CHECK(hipGetLastError());
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            char what[128];
            snprintf(what, sizeof what, "LDS table of %d records (%zu KiB), %2d lanes per record", M, lds_bytes / 1024, g);
            report(what, ms);
        }
        for (double share : {0.5, 0.67, 0.8}) {
            fill(M, share);
            CHECK(hipMemcpy(d_t, tab.data(), size_t(n_rec) * sizeof(Rec), hipMemcpyHostToDevice));
            starts(1, n_rec);
            CHECK(hipMemcpy(d_start, start.data(), lanes * sizeof(int), hipMemcpyHostToDevice));
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_chase_mixed<1024>, dim3(lanes / 1024), dim3(1024), lds_bytes, 0, d_t, M, d_start, iters, d_out, d_clk);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
            }
            CHECK(hipGetLastError());
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            char what[128];
            snprintf(what, sizeof what, "mixed, table of %d records, %.0f %% of the visits in LDS", M, 100.0 * share);
            report(what, ms);
        }
    }
    return 0;
}
