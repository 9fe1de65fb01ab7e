// microbench_graph.hip -- what does a chain of N small dependent kernels cost when launched one by one
// on a stream, and when replayed as a captured hipGraph?  (The device-side octree build is ~20 kernels
// of ~5 us.)  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_graph.hip -o /tmp/mb_graph
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

__global__ void k_small(float* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}

int main() {
    const int n = 65536, chain = 20, reps = 200;
    float* p;
    CHECK(hipMalloc(&p, n * sizeof(float)));
    CHECK(hipMemset(p, 0, n * sizeof(float)));
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto run_stream = [&] { for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, s, p, n); };
    for (int w = 0; w < 20; ++w) run_stream();
    CHECK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) run_stream();
    CHECK(hipStreamSynchronize(s));
    double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;

    hipGraph_t graph; hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    run_stream();
    CHECK(hipStreamEndCapture(s, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int w = 0; w < 20; ++w) CHECK(hipGraphLaunch(exec, s));
    CHECK(hipStreamSynchronize(s));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(exec, s));
    CHECK(hipStreamSynchronize(s));
    double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    // with a host synchronisation after every chain (as a Barnes-Hut step with the device build has)
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) { run_stream(); CHECK(hipStreamSynchronize(s)); }
    double us_stream_sync = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) { CHECK(hipGraphLaunch(exec, s)); CHECK(hipStreamSynchronize(s)); }
    double us_graph_sync = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("chain of %d dependent small kernels: stream %.1f us (%.2f per kernel), graph %.1f us (%.2f per kernel)\n", chain,
           us_stream, us_stream / chain, us_graph, us_graph / chain);
    printf("  with a host sync after every chain: stream %.1f us, graph %.1f us\n", us_stream_sync, us_graph_sync);
    return 0;
}
