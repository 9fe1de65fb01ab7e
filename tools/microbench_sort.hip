// Which rocPRIM radix-sort configuration is fastest for the device build's sort: (u64 key, i32 id) pairs, key bits [15, 63),
// 65 536 .. 4 M items?   hipcc -O3 --offload-arch=gfx950 -o build/microbench_sort tools/microbench_sort.hip && build/microbench_sort
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <random>
#include <vector>

template <class Config>
static float run(const char* name, size_t n, const unsigned long long* kin, unsigned long long* kout, const int* vin, int* vout, int begin_bit) {
    size_t tb = 0;
    if (rocprim::radix_sort_pairs<Config>(nullptr, tb, kin, kout, vin, vout, n, begin_bit, 63, nullptr) != hipSuccess) { printf("%s: size query failed\n", name); return -1; }
    void* tmp = nullptr;
    hipMalloc(&tmp, tb);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 5; ++w) rocprim::radix_sort_pairs<Config>(tmp, tb, kin, kout, vin, vout, n, begin_bit, 63, nullptr);
    hipEventRecord(a, nullptr);
    const int reps = 50;
    for (int r = 0; r < reps; ++r) rocprim::radix_sort_pairs<Config>(tmp, tb, kin, kout, vin, vout, n, begin_bit, 63, nullptr);
    hipEventRecord(b, nullptr);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipFree(tmp);
    printf("  %-34s n=%8zu bits [%d,63): %8.2f us per sort\n", name, n, begin_bit, 1e3f * ms / reps);
    return ms;
}

using Ks = rocprim::kernel_config<256, 12>;
template <unsigned B> using OS = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::radix_sort_onesweep_config<Ks, Ks, B>, 0>;
template <unsigned B, unsigned T, unsigned I> using OS2 = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::radix_sort_onesweep_config<Ks, rocprim::kernel_config<T, I>, B>, 0>;

int main() {
    for (size_t n : {size_t(65536), size_t(100001), size_t(524288), size_t(4194304)}) {
        std::vector<unsigned long long> hk(n);
        std::vector<int> hv(n);
        std::mt19937_64 rng(7);
        for (size_t i = 0; i < n; ++i) { hk[i] = rng() >> 1; hv[i] = int(i); }
        unsigned long long *kin, *kout; int *vin, *vout;
        hipMalloc(&kin, n * 8); hipMalloc(&kout, n * 8); hipMalloc(&vin, n * 4); hipMalloc(&vout, n * 4);
        hipMemcpy(kin, hk.data(), n * 8, hipMemcpyHostToDevice);
        hipMemcpy(vin, hv.data(), n * 4, hipMemcpyHostToDevice);
        printf("n = %zu\n", n);
        run<rocprim::default_config>("default", n, kin, kout, vin, vout, 15);
        run<rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>>("tuned onesweep (merge limit 0)", n, kin, kout, vin, vout, 15);
        run<rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>>("tuned onesweep, 32 bits", n, kin, kout, vin, vout, 31);
        run<OS<8>>("onesweep 8 bits", n, kin, kout, vin, vout, 15);
        run<OS<6>>("onesweep 6 bits", n, kin, kout, vin, vout, 15);
        run<OS<7>>("onesweep 7 bits", n, kin, kout, vin, vout, 15);
        run<OS2<8, 256, 8>>("onesweep 8 bits, 256 x 8", n, kin, kout, vin, vout, 15);
        run<OS2<8, 256, 4>>("onesweep 8 bits, 256 x 4", n, kin, kout, vin, vout, 15);
        run<OS2<8, 128, 8>>("onesweep 8 bits, 128 x 8", n, kin, kout, vin, vout, 15);
        run<rocprim::default_config>("default, 40 bits", n, kin, kout, vin, vout, 23);
        run<rocprim::default_config>("default, 32 bits", n, kin, kout, vin, vout, 31);
        hipFree(kin); hipFree(kout); hipFree(vin); hipFree(vout);
    }
    return 0;
}
