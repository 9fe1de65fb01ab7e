// ipc_probe.hip -- does this stack give two PROCESSES on ONE GPU what an in-library transport needs?
//   A. hipIpcGetMemHandle / hipIpcOpenMemHandle between processes on the same device
//   B. interprocess events (hipEventInterprocess + hipIpcGetEventHandle / hipIpcOpenEventHandle) as stream-ordered
//      "data ready" / "buffer free" signals
//   C. the same signals as flags in host shared memory (hipHostRegister'ed POSIX shm), set and polled by one-lane
//      kernels (the poll is bounded in time: every wave reaches its exit)
// The parent never touches HIP: it creates the shared-memory file and starts the ranks as fresh processes.
// Build: hipcc -O2 --offload-arch=gfx950 tools/ipc_probe.hip -o build/ipc_probe -lrt -pthread
// Run:   build/ipc_probe [ranks=2] [rounds=200]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/wait.h>
#include <thread>
#include <unistd.h>
#include <vector>

constexpr int kMaxRanks = 8;
struct Ctrl {
    std::atomic<int> barrier[64];
    hipIpcMemHandle_t mem[kMaxRanks];
    hipIpcEventHandle_t ev_ready[kMaxRanks], ev_done[kMaxRanks];
    std::atomic<int> posted[kMaxRanks], consumed[kMaxRanks];
    // mode C: flags the GPU reads and writes
    alignas(64) volatile int ready[kMaxRanks * 16];
    alignas(64) volatile int done[kMaxRanks * 16];
    std::atomic<int> failed;
};

#define CK(expr)                                                                                         \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            std::fprintf(stderr, "[rank %d] %s -> %s\n", g_rank, #expr, hipGetErrorString(e_));          \
            return 10;                                                                                   \
        }                                                                                                \
    } while (0)

static int g_rank = -1;
using clk = std::chrono::steady_clock;

static bool host_wait(std::atomic<int>& a, int at_least, double seconds = 20.0) {
    const auto t0 = clk::now();
    while (a.load(std::memory_order_acquire) < at_least) {
        if (std::chrono::duration<double>(clk::now() - t0).count() > seconds) return false;
        std::this_thread::yield();
    }
    return true;
}
static bool barrier(Ctrl* c, int id, int world) {
    c->barrier[id].fetch_add(1, std::memory_order_acq_rel);
    return host_wait(c->barrier[id], world);
}

__global__ void k_delay_fill(int* buf, int n, int value, long long spin) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < spin) {}
    }
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = value;
}
__global__ void k_check(const int* buf, int n, int value, int* errors) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (buf[i] != value) {
            atomicAdd(errors, 1);
            const int d = buf[i] - value;   // stale (< 0), overwritten early (> 0), neither
            atomicAdd(errors + (d == -1 ? 1 : d == 1 ? 2 : d < 0 ? 3 : 4), 1);
            atomicMin(errors + 5, value);   // first round that went wrong
            atomicMax(errors + 6, value);
        }
}
__global__ void k_copy(int* dst, const int* src, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void k_set_flag(volatile int* flag, int value) {
    __threadfence_system();
    __hip_atomic_store(const_cast<int*>(flag), value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// bounded: ~2 s of the 100 MHz wall clock, then the error word is raised and the wave leaves
__global__ void k_wait_flag(volatile int* flag, int at_least, int* errors) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(const_cast<int*>(flag), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < at_least) {
        if (wall_clock64() - t0 > 200000000LL) { atomicAdd(errors, 1000000); return; }
        __builtin_amdgcn_s_sleep(32);
    }
    __threadfence_system();
}

static int child(const char* shm_name, int rank, int world, int rounds) {
    g_rank = rank;
    int fd = shm_open(shm_name, O_RDWR, 0600);
    if (fd < 0) { std::perror("shm_open"); return 2; }
    Ctrl* c = static_cast<Ctrl*>(mmap(nullptr, sizeof(Ctrl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
    if (c == MAP_FAILED) { std::perror("mmap"); return 2; }
    CK(hipSetDevice(0));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int n = 1 << 18;   // 1 MiB of ints
    int *win = nullptr, *dst = nullptr, *errors = nullptr;
    CK(hipMalloc(&win, n * sizeof(int)));
    CK(hipMalloc(&dst, size_t(n) * world * sizeof(int)));
    CK(hipMalloc(&errors, 8 * sizeof(int)));
    CK(hipMemset(errors, 0, 8 * sizeof(int)));
    // ---- A: memory handles
    CK(hipIpcGetMemHandle(&c->mem[rank], win));
    if (!barrier(c, 0, world)) { std::fprintf(stderr, "[rank %d] barrier 0 timed out\n", rank); return 3; }
    int* peer_win[kMaxRanks] = {};
    for (int r = 0; r < world; ++r) {
        if (r == rank) { peer_win[r] = win; continue; }
        CK(hipIpcOpenMemHandle(reinterpret_cast<void**>(&peer_win[r]), c->mem[r], hipIpcMemLazyEnablePeerAccess));
    }
    std::printf("[rank %d] A ok: opened %d peer windows\n", rank, world - 1);
    // ---- B: interprocess events
    bool ev_ok = true;
    hipEvent_t my_ready = nullptr, my_done = nullptr, peer_ready[kMaxRanks] = {}, peer_done[kMaxRanks] = {};
    {
        hipError_t e = hipEventCreateWithFlags(&my_ready, hipEventInterprocess | hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&my_done, hipEventInterprocess | hipEventDisableTiming);
        if (e == hipSuccess) e = hipIpcGetEventHandle(&c->ev_ready[rank], my_ready);
        if (e == hipSuccess) e = hipIpcGetEventHandle(&c->ev_done[rank], my_done);
        if (e != hipSuccess) { std::printf("[rank %d] B: interprocess event create/export failed: %s\n", rank, hipGetErrorString(e)); ev_ok = false; c->failed.store(1); }
    }
    if (!barrier(c, 1, world)) return 3;
    if (c->failed.load()) ev_ok = false;
    if (ev_ok)
        for (int r = 0; r < world && ev_ok; ++r) {
            if (r == rank) continue;
            hipError_t e = hipIpcOpenEventHandle(&peer_ready[r], c->ev_ready[r]);
            if (e == hipSuccess) e = hipIpcOpenEventHandle(&peer_done[r], c->ev_done[r]);
            if (e != hipSuccess) { std::printf("[rank %d] B: open event handle failed: %s\n", rank, hipGetErrorString(e)); ev_ok = false; c->failed.store(1); }
        }
    if (!barrier(c, 2, world)) return 3;
    if (c->failed.load()) ev_ok = false;
    auto run_b = [&](hipStream_t st) -> int {
        const auto t0 = clk::now();
        for (int round = 1; round <= rounds; ++round) {
            // my window may be overwritten once every peer has read the previous round
            if (round > 1)
                for (int r = 0; r < world; ++r) {
                    if (r == rank) continue;
                    if (!host_wait(c->consumed[r], round - 1)) { std::fprintf(stderr, "[rank %d] B consumed wait timed out\n", rank); return 4; }
                    CK(hipStreamWaitEvent(st, peer_done[r], 0));
                }
            // a slow producer on odd rounds on rank 0 / even rounds elsewhere: a missing wait reads stale data
            const long long spin = ((round + rank) & 1) ? 50000 : 0;   // 0.5 ms
            k_delay_fill<<<64, 256, 0, st>>>(win, n, rank * 100000 + round, spin);
            CK(hipEventRecord(my_ready, st));
            c->posted[rank].store(round, std::memory_order_release);
            for (int r = 0; r < world; ++r) {
                if (r == rank) continue;
                if (!host_wait(c->posted[r], round)) { std::fprintf(stderr, "[rank %d] B posted wait timed out\n", rank); return 4; }
                CK(hipStreamWaitEvent(st, peer_ready[r], 0));
                CK(hipMemcpyAsync(dst + size_t(r) * n, peer_win[r], n * sizeof(int), hipMemcpyDeviceToDevice, st));
                k_check<<<64, 256, 0, st>>>(dst + size_t(r) * n, n, r * 100000 + round, errors);
            }
            CK(hipEventRecord(my_done, st));
            c->consumed[rank].store(round, std::memory_order_release);
        }
        CK(hipStreamSynchronize(st));
        const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
        int herr = 0;
        CK(hipMemcpy(&herr, errors, sizeof(int), hipMemcpyDeviceToHost));
        std::printf("[rank %d] B interprocess events: %d rounds, %d wrong words, %.3f ms per round\n", rank, rounds, herr, ms / rounds);
        CK(hipMemset(errors, 0, sizeof(int)));
        return 0;
    };
    if (ev_ok) {
        // what does the opened event answer to the plain calls?
        for (int r = 0; r < world; ++r) {
            if (r == rank) continue;
            std::printf("[rank %d] B: hipEventQuery(opened) = %s, hipStreamWaitEvent(null stream) = %s, (created stream) = %s\n", rank,
                        hipGetErrorString(hipEventQuery(peer_ready[r])), hipGetErrorString(hipStreamWaitEvent(nullptr, peer_ready[r], 0)),
                        hipGetErrorString(hipStreamWaitEvent(st, peer_ready[r], 0)));
        }
        (void)hipGetLastError();
        hipStream_t st_blocking = nullptr;
        CK(hipStreamCreate(&st_blocking));
        int rb = run_b(st_blocking);
        std::printf("[rank %d] B on a blocking stream: rc %d\n", rank, rb);
        (void)hipGetLastError();
        c->posted[rank].store(1 << 30); c->consumed[rank].store(1 << 30);   // (let a peer that got further fall through)
    }
    if (!barrier(c, 3, world)) return 3;
    // ---- C: flags in registered host shared memory, set / polled by kernels
    {
        hipError_t e = hipHostRegister(c, sizeof(Ctrl), hipHostRegisterMapped);
        if (e != hipSuccess) { std::printf("[rank %d] C: hipHostRegister failed: %s\n", rank, hipGetErrorString(e)); return 0; }
        Ctrl* dc = nullptr;
        CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&dc), c, 0));
        if (!barrier(c, 4, world)) return 3;
        for (int variant = 0; variant < 3; ++variant) {
        const int base = variant * rounds;
        int init[8] = {0, 0, 0, 0, 0, 0x7fffffff, 0, 0};
        CK(hipMemcpy(errors, init, sizeof init, hipMemcpyHostToDevice));
        const auto t0 = clk::now();
        for (int round = base + 1; round <= base + rounds; ++round) {
            if (round > 1)
                for (int r = 0; r < world; ++r)
                    if (r != rank) k_wait_flag<<<1, 1, 0, st>>>(&dc->done[r * 16], round - 1, errors);
            const long long spin = ((round + rank) & 1) ? 50000 : 0;
            k_delay_fill<<<64, 256, 0, st>>>(win, n, rank * 100000 + 50000 + round, spin);
            k_set_flag<<<1, 1, 0, st>>>(&dc->ready[rank * 16], round);
            for (int r = 0; r < world; ++r) {
                if (r == rank) continue;
                k_wait_flag<<<1, 1, 0, st>>>(&dc->ready[r * 16], round, errors);
                if (variant == 0) CK(hipMemcpyAsync(dst + size_t(r) * n, peer_win[r], n * sizeof(int), hipMemcpyDeviceToDevice, st));
                else if (variant == 1) k_copy<<<64, 256, 0, st>>>(dst + size_t(r) * n, peer_win[r], n);
                if (variant == 2) k_check<<<64, 256, 0, st>>>(peer_win[r], n, r * 100000 + 50000 + round, errors);   // read in place
                else k_check<<<64, 256, 0, st>>>(dst + size_t(r) * n, n, r * 100000 + 50000 + round, errors);
            }
            k_set_flag<<<1, 1, 0, st>>>(&dc->done[rank * 16], round);
        }
        CK(hipStreamSynchronize(st));
        const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
        int herr[8];
        CK(hipMemcpy(herr, errors, sizeof herr, hipMemcpyDeviceToHost));
        std::printf("[rank %d] C%d (%s): %d rounds, wrong words %d (>= 1000000: a poll timed out) [stale-by-1 %d, early-by-1 %d, older %d, newer %d; first/last wrong value %d/%d], %.3f ms per round\n",
                    rank, variant, variant == 0 ? "hipMemcpyAsync" : variant == 1 ? "copy kernel" : "read in place", rounds, herr[0], herr[1], herr[2], herr[3], herr[4], herr[5], herr[6], ms / rounds);
        }
        if (!barrier(c, 5, world)) return 3;
        (void)hipHostUnregister(c);
    }
    for (int r = 0; r < world; ++r) if (r != rank && peer_win[r]) (void)hipIpcCloseMemHandle(peer_win[r]);
    std::fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 6 && std::strcmp(argv[1], "child") == 0) return child(argv[2], std::atoi(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]));
    const int world = argc > 1 ? std::atoi(argv[1]) : 2;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 200;
    if (world < 2 || world > kMaxRanks) return 1;
    char name[64];
    std::snprintf(name, sizeof name, "/nbody_ipc_probe_%d", int(getpid()));
    int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Ctrl)) != 0) { std::perror("shm"); return 1; }
    void* p = mmap(nullptr, sizeof(Ctrl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    std::memset(p, 0, sizeof(Ctrl));
    std::vector<pid_t> kids;
    for (int r = 0; r < world; ++r) {
        pid_t pid = fork();
        if (pid == 0) {   // (this process has never touched HIP)
            std::string rs = std::to_string(r), ws = std::to_string(world), ks = std::to_string(rounds);
            execl(argv[0], argv[0], "child", name, rs.c_str(), ws.c_str(), ks.c_str(), (char*)nullptr);
            _exit(127);
        }
        kids.push_back(pid);
    }
    int bad = 0;
    for (pid_t k : kids) {
        int st = 0;
        waitpid(k, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { std::fprintf(stderr, "rank process %d ended with status %d\n", int(k), st); bad = 1; }
    }
    shm_unlink(name);
    return bad;
}
