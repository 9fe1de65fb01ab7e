// transport_ipc.hip -- the exchanges between ranks that share one device (G processes, or threads of one process, on
// ONE GPU): what lets the production multi-rank step run on a one-GPU box (RCCL refuses two ranks on one device).
//
// Shape of one grouped exchange on rank `me` (everything stream-ordered, nothing waits for the device):
//   host   publish one descriptor {seq, bytes, offset, window generation} per outgoing message (ring in shared memory)
//   stream wait until the receivers of the window's previous contents are done with it        (k_ipc_wait, one lane per flag)
//          stage the outgoing payloads into the own window                                      (k_ipc_copy, one launch)
//          raise ready[me -> r] = seq for every destination r                                   (k_ipc_signal)
//   host   read the descriptor of every incoming message (waits for the SENDER'S HOST to have published it -- the
//          rendezvous ncclGroupEnd also is -- never for a device); byte counts must agree or the run is aborted loudly
//   stream wait for ready[r -> me] = seq of every source r                                      (k_ipc_wait)
//          copy out of the sources' windows (opened with hipIpcOpenMemHandle) into the destinations  (k_ipc_copy)
//          raise done[r -> me] = seq                                                            (k_ipc_signal)
// A send is complete for the caller once its payload is staged (a buffered send: the user buffer may be reused by the
// next kernel on the stream, as after ncclSend).  Flags live in host shared memory (POSIX shm, hipHostRegister'ed by
// every rank); they are sequence numbers per ordered pair of ranks, so a rank that issues its operations in another
// order than its peers is found out (a wait that runs out of time, or a byte-count mismatch) instead of corrupting data.
// Every device-side poll is bounded (NBODY_COMM_TIMEOUT_S, default 30 s of the wall clock, and the shared abort word):
// every wave reaches its exit, a dead peer cannot hang the GPU.
// Measured first with tools/ipc_probe.hip (profiles/r03_ipc_probe.txt): memory handles work between processes on one
// device; interprocess EVENTS do not (hipStreamWaitEvent on an opened handle: invalid argument), hence the flags.
// Ranks may also be THREADS of one process (the tests run 8 ranks as 4 x 2: a GPU box admits six GPU processes); then
// every stream needs its own hardware queue (GPU_MAX_HW_QUEUES >= 2 per rank + 2): streams folded onto one queue run in
// submission order, and a device-side wait of one rank would sit in front of the kernel of the other it waits for.
// Windows and mappings outlive the transports (struct Arena below): on this driver, unmapping a peer's window
// (hipIpcCloseMemHandle) or freeing an exported one, with other ranks' queues busy then or soon after, froze every process on
// the device for 30-60 s (profiles/r03_ipc_close_stall.txt).  So neither ever happens: an exported window goes back to a
// process-wide pool when its transport is done with it and serves the next transport of the process; a mapped window stays
// mapped, and is found again by its handle.  The process's exit takes them all.
#include "transport.h"
#include "../../include/nbody_hip.h"

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <mutex>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace nbody {

namespace {

constexpr int kMaxRanks = 16;
constexpr int kRing = 64;              // descriptors in flight per ordered pair of ranks
constexpr uint64_t kMagic = 0x314350494e42ull;   // "NBIPC1"
constexpr char kIdTag[8] = {'N', 'B', 'I', 'P', 'C', '1', 0, 0};

struct Desc { uint64_t seq, bytes, offset, gen; };
struct alignas(64) Flag { unsigned int v; };   // one cache line each: the device polls these
struct RankInfo {
    std::atomic<int> present;
    int pid;
    int device;
    std::atomic<uint64_t> win_gen;    // 0: no window yet
    hipIpcMemHandle_t win_handle;
    uint64_t win_raw;                 // the window's address in its owner's process (ranks of the same process use it)
    uint64_t win_bytes;
    char blob[256];                   // host_all_gather
};
struct Ctrl {
    uint64_t magic;
    std::atomic<int> world, arrived, bar_count, bar_gen;
    RankInfo rank[kMaxRanks];
    std::atomic<uint64_t> posted[kMaxRanks][kMaxRanks];    // [src][dst] descriptors the source's host has published
    std::atomic<uint64_t> consumed[kMaxRanks][kMaxRanks];  // [src][dst] descriptors the destination's host has read
    Desc ring[kMaxRanks][kMaxRanks][kRing];
    Flag ready[kMaxRanks][kMaxRanks];   // [src][dst] device: message seq of src is staged
    Flag done[kMaxRanks][kMaxRanks];    // [src][dst] device: dst has copied message seq out of src's window
    Flag abort_word;                    // anybody, host or device: give up
    Flag dev_timeout[kMaxRanks];        // a device-side wait of this rank ran out of time
};

struct FlagList {
    int n;
    unsigned int* flag[kMaxRanks];
    unsigned int value[kMaxRanks];
};

__global__ void k_ipc_signal(FlagList l) {
    __threadfence_system();
    const int i = threadIdx.x;
    if (i < l.n) __hip_atomic_store(l.flag[i], l.value[i], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one lane per flag; sequence numbers are compared modulo 2^32.  Bounded: `ticks` of the constant-rate wall clock.
__global__ void k_ipc_wait(FlagList l, unsigned int* abort_word, unsigned int* timeout_word, long long ticks) {
    const int i = threadIdx.x;
    if (i < l.n) {
        const long long t0 = wall_clock64();
        while (int(__hip_atomic_load(l.flag[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - l.value[i]) < 0) {
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
            if (wall_clock64() - t0 > ticks) {
                __hip_atomic_store(timeout_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(64);
        }
    }
    __threadfence_system();
}

// the payload copies of one phase of a group (staging, or copying out of the peers' windows) in one launch: grid.y = the
// copy, grid.x strides over it.  Plain kernels on the caller's stream rather than hipMemcpyAsync: an SDMA engine is a
// queue shared by all streams (and rank threads) of a process, and a copy parked there behind a device-side wait of
// another rank of the same process is a deadlock the runtime cannot see.
constexpr int kMaxCopies = 40;
struct CopyList {
    int n;
    struct { char* dst; const char* src; size_t bytes; } c[kMaxCopies];
};
__global__ void k_ipc_copy(CopyList l) {
    char* dst = l.c[blockIdx.y].dst;
    const char* src = l.c[blockIdx.y].src;
    const size_t bytes = l.c[blockIdx.y].bytes;
    const size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x, nth = size_t(gridDim.x) * blockDim.x;
    if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | bytes) & 15) == 0) {
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (size_t i = tid; i < bytes / 16; i += nth) d4[i] = s4[i];
    } else if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | bytes) & 3) == 0) {
        const unsigned int* s1 = reinterpret_cast<const unsigned int*>(src);
        unsigned int* d1 = reinterpret_cast<unsigned int*>(dst);
        for (size_t i = tid; i < bytes / 4; i += nth) d1[i] = s1[i];
    } else {
        for (size_t i = tid; i < bytes; i += nth) dst[i] = src[i];
    }
}

// ---- process-wide: exported windows (reused, never freed) and mapped peer windows (never unmapped); see the header
struct Arena {
    struct Win { char* p; size_t bytes; int device; hipIpcMemHandle_t handle; bool busy; };
    struct Map { hipIpcMemHandle_t handle; int device; char* p; };
    std::mutex m;
    std::vector<Win> wins;
    std::vector<Map> maps;

    // a window of at least `bytes` on `device` (the smallest idle one that fits, else a new allocation)
    hipError_t acquire(int device, size_t bytes, char** p, size_t* got, hipIpcMemHandle_t* handle) {
        std::lock_guard<std::mutex> lock(m);
        Win* best = nullptr;
        for (Win& w : wins)
            if (!w.busy && w.device == device && w.bytes >= bytes && (!best || w.bytes < best->bytes)) best = &w;
        if (!best) {
            void* fresh = nullptr;
            hipError_t e = hipMalloc(&fresh, bytes);
            if (e != hipSuccess) return e;
            Win w{static_cast<char*>(fresh), bytes, device, {}, false};
            e = hipIpcGetMemHandle(&w.handle, fresh);
            if (e != hipSuccess) { (void)hipFree(fresh); return e; }   // (never exported: freeing it is harmless)
            wins.push_back(w);
            best = &wins.back();
        }
        best->busy = true;
        *p = best->p; *got = best->bytes; *handle = best->handle;
        return hipSuccess;
    }
    void release(char* p) {
        std::lock_guard<std::mutex> lock(m);
        for (Win& w : wins) if (w.p == p) w.busy = false;
    }
    hipError_t open(int device, const hipIpcMemHandle_t& handle, char** p) {
        std::lock_guard<std::mutex> lock(m);
        for (const Map& k : maps)
            if (k.device == device && std::memcmp(&k.handle, &handle, sizeof(handle)) == 0) { *p = k.p; return hipSuccess; }
        void* q = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&q, handle, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return e;
        maps.push_back(Map{handle, device, static_cast<char*>(q)});
        *p = static_cast<char*>(q);
        return hipSuccess;
    }
};
Arena& arena() {
    static Arena* a = new Arena;   // (never destroyed: transports of static-lifetime handles may still release into it at exit)
    return *a;
}

using clk = std::chrono::steady_clock;

double timeout_seconds() {
    if (const char* v = std::getenv("NBODY_COMM_TIMEOUT_S")) { const double t = std::atof(v); if (t > 0) return t; }
    return 30.0;
}

size_t ctrl_bytes() {
    const size_t page = size_t(sysconf(_SC_PAGESIZE));
    return (sizeof(Ctrl) + page - 1) / page * page;
}

struct Op { int kind; void* p; size_t bytes; int peer; };   // 0 send, 1 recv, 2 all-gather

class IpcTransport final : public Transport {
public:
    IpcTransport(int rank, int world, int device) : rank_(rank), world_(world), device_(device) {}

    int open(const char* shm_name) {
        timeout_s_ = timeout_seconds();
        int fd = shm_open(shm_name, O_RDWR, 0600);
        if (fd < 0) return fail(NBODY_ERR_COMM, std::string("ipc transport: shm_open(") + shm_name + ") failed (is the id from nbody_comm_local_id of a process on this host?)");
        void* p = mmap(nullptr, ctrl_bytes(), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        ::close(fd);
        if (p == MAP_FAILED) return fail(NBODY_ERR_COMM, "ipc transport: mmap of the control block failed");
        c_ = static_cast<Ctrl*>(p);
        if (c_->magic != kMagic) return fail(NBODY_ERR_COMM, "ipc transport: the shared block does not carry this library's tag");
        int expect = 0;
        if (!c_->world.compare_exchange_strong(expect, world_) && expect != world_)
            return fail(NBODY_ERR_COMM, "ipc transport: ranks disagree on world_size (" + std::to_string(expect) + " vs " + std::to_string(world_) + ")");
        RankInfo& me = c_->rank[rank_];
        int was = 0;
        if (!me.present.compare_exchange_strong(was, 1)) return fail(NBODY_ERR_COMM, "ipc transport: rank " + std::to_string(rank_) + " joined twice");
        me.pid = int(getpid());
        me.device = device_;
        c_->arrived.fetch_add(1, std::memory_order_acq_rel);
        if (!host_wait([&] { return c_->arrived.load(std::memory_order_acquire) >= world_; }))
            return fail(NBODY_ERR_COMM, "ipc transport: only " + std::to_string(c_->arrived.load()) + " of " + std::to_string(world_) + " ranks joined within " +
                                            std::to_string(int(timeout_s_)) + " s");
        joined_ = true;
        if (rank_ == 0) (void)shm_unlink(shm_name);   // everybody has it mapped: the name can go
        if (hipHostRegister(c_, ctrl_bytes(), hipHostRegisterMapped) != hipSuccess) return fail(NBODY_ERR_HIP, "ipc transport: hipHostRegister of the control block failed");
        registered_ = true;
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&dc_), c_, 0) != hipSuccess) return fail(NBODY_ERR_HIP, "ipc transport: hipHostGetDevicePointer failed");
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device_) != hipSuccess || khz <= 0) khz = 100000;
        ticks_ = (long long)(timeout_s_ * 1e3 * double(khz));
        return NBODY_OK;
    }

    ~IpcTransport() override {
        if (c_ && joined_) {
            (void)hipDeviceSynchronize();
            // the own window serves another transport only when every receiver is done with what was sent (a dead peer: a short wait)
            const double keep = timeout_s_;
            timeout_s_ = std::min(timeout_s_, 5.0);
            for (int r = 0; r < world_; ++r)
                if (r != rank_ && sent_[r] > 0) (void)host_wait([&] { return int(flag_load(c_->done[rank_][r]) - (unsigned int)sent_[r]) >= 0; });
            timeout_s_ = keep;
        }
        if (win_) arena().release(win_);
        if (registered_) (void)hipHostUnregister(c_);
        if (c_) (void)munmap(c_, ctrl_bytes());
    }

    const char* name() const override { return "ipc"; }
    int rank() const override { return rank_; }
    int world() const override { return world_; }

    int all_gather(void* buf, size_t bytes, hipStream_t s) override { return add(Op{2, buf, bytes, -1}, s); }
    int send(const void* p, size_t bytes, int peer, hipStream_t s) override {
        if (peer < 0 || peer >= world_ || peer == rank_) return fail(NBODY_ERR_INVALID, "ipc transport: bad send peer");
        return add(Op{0, const_cast<void*>(p), bytes, peer}, s);
    }
    int recv(void* p, size_t bytes, int peer, hipStream_t s) override {
        if (peer < 0 || peer >= world_ || peer == rank_) return fail(NBODY_ERR_INVALID, "ipc transport: bad recv peer");
        return add(Op{1, p, bytes, peer}, s);
    }
    int group_begin() override { ++depth_; return NBODY_OK; }
    int group_end() override {
        if (depth_ <= 0) return fail(NBODY_ERR_INVALID, "ipc transport: group_end without group_begin");
        if (--depth_ > 0) return NBODY_OK;
        return run();
    }

    int host_all_gather(const void* mine, void* all, size_t bytes) override {
        if (bytes > sizeof(c_->rank[0].blob)) return fail(NBODY_ERR_INVALID, "host_all_gather: blob larger than 256 bytes");
        std::memcpy(c_->rank[rank_].blob, mine, bytes);
        int rc = barrier();
        if (rc) return rc;
        for (int r = 0; r < world_; ++r) std::memcpy(static_cast<char*>(all) + size_t(r) * bytes, c_->rank[r].blob, bytes);
        return barrier();   // (nobody rewrites its blob before everybody has read it)
    }

    int check() override {
        if (flag_load(c_->dev_timeout[rank_]))
            return fail(NBODY_ERR_COMM, "ipc transport: a device-side wait of rank " + std::to_string(rank_) + " ran out of time (" + std::to_string(int(timeout_s_)) +
                                            " s): a peer never sent, or never took, a message this rank's step expects");
        if (flag_load(c_->abort_word)) return fail(NBODY_ERR_COMM, "ipc transport: another rank gave up (its own error has the reason)");
        return NBODY_OK;
    }

private:
    int fail(int code, const std::string& msg) {
        err_ = msg;
        if (c_ && joined_ && code == NBODY_ERR_COMM) flag_store(c_->abort_word, 1u);   // the peers should not wait out their timeouts
        return code;
    }
    static unsigned int flag_load(const Flag& f) { return __atomic_load_n(&f.v, __ATOMIC_ACQUIRE); }
    static void flag_store(Flag& f, unsigned int v) { __atomic_store_n(&f.v, v, __ATOMIC_RELEASE); }

    template <class Pred> bool host_wait(Pred ok) {
        const auto t0 = clk::now();
        for (unsigned spin = 0;; ++spin) {
            if (ok()) return true;
            if (joined_ && flag_load(c_->abort_word)) return false;
            if (std::chrono::duration<double>(clk::now() - t0).count() > timeout_s_) return false;
            if (spin < 2000) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }

    int barrier() {
        const int gen = c_->bar_gen.load(std::memory_order_acquire);
        if (c_->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == world_) {
            c_->bar_count.store(0, std::memory_order_relaxed);
            c_->bar_gen.store(gen + 1, std::memory_order_release);
            return NBODY_OK;
        }
        if (!host_wait([&] { return c_->bar_gen.load(std::memory_order_acquire) != gen; }))
            return fail(NBODY_ERR_COMM, "ipc transport: a rank did not reach the barrier within " + std::to_string(int(timeout_s_)) + " s");
        return NBODY_OK;
    }

    int add(const Op& op, hipStream_t s) {
        if (ops_.empty()) stream_ = s;
        else if (s != stream_) return fail(NBODY_ERR_INVALID, "ipc transport: the operations of one group must share a stream");
        ops_.push_back(op);
        if (depth_ == 0) return run();
        return NBODY_OK;
    }

    static unsigned copy_blocks(size_t bytes) { return unsigned(std::max<size_t>(1, std::min<size_t>(512, (bytes + 16 * 256 - 1) / (16 * 256)))); }

    int ensure_window(size_t need) {
        if (need <= win_bytes_) return NBODY_OK;
        // what was sent so far has to be out of the old window before it is retired
        for (int r = 0; r < world_; ++r)
            if (r != rank_ && sent_[r] > 0 && !host_wait([&] { return int(flag_load(c_->done[rank_][r]) - (unsigned int)sent_[r]) >= 0; }))
                return fail(NBODY_ERR_COMM, "ipc transport: rank " + std::to_string(r) + " never took message " + std::to_string(sent_[r]) + " of rank " + std::to_string(rank_));
        size_t bytes = std::max<size_t>(need + need / 2, size_t(1) << 20);
        bytes = (bytes + (size_t(2) << 20) - 1) / (size_t(2) << 20) * (size_t(2) << 20);
        char* fresh = nullptr;
        RankInfo& me = c_->rank[rank_];
        hipIpcMemHandle_t handle;
        if (arena().acquire(device_, bytes, &fresh, &bytes, &handle) != hipSuccess) return fail(NBODY_ERR_HIP, "ipc transport: no memory for the window (hipMalloc / hipIpcGetMemHandle failed)");
        me.win_handle = handle;
        if (win_) arena().release(win_);   // (everything sent from it has been taken: see the wait above)
        win_ = fresh;
        win_bytes_ = bytes;
        me.win_raw = uint64_t(reinterpret_cast<uintptr_t>(fresh));
        me.win_bytes = bytes;
        me.win_gen.store(++win_gen_, std::memory_order_release);
        return NBODY_OK;
    }

    int map_window(int r, uint64_t gen, char** out) {
        if (peer_gen_[r] == gen && peer_ptr_[r]) { *out = peer_ptr_[r]; return NBODY_OK; }
        RankInfo& pi = c_->rank[r];
        if (pi.win_gen.load(std::memory_order_acquire) != gen) return fail(NBODY_ERR_COMM, "ipc transport: window generation of rank " + std::to_string(r) + " moved under a message");
        if (pi.pid == int(getpid())) {
            peer_ptr_[r] = reinterpret_cast<char*>(uintptr_t(pi.win_raw));
        } else {
            const hipIpcMemHandle_t handle = pi.win_handle;
            const hipError_t e = arena().open(device_, handle, &peer_ptr_[r]);   // (the mapping of the window it replaces stays: see the header)
            if (e != hipSuccess) { peer_ptr_[r] = nullptr; return fail(NBODY_ERR_HIP, std::string("ipc transport: hipIpcOpenMemHandle: ") + hipGetErrorString(e)); }
        }
        peer_gen_[r] = gen;
        *out = peer_ptr_[r];
        return NBODY_OK;
    }

    int run() {
        std::vector<Op> ops;
        ops.swap(ops_);
        hipStream_t s = stream_;
        int rc = check();
        if (rc) return rc;
        if (world_ == 1) {   // (all-gathers of a world of one: the own part is in place; there is nobody to send to)
            for (const Op& o : ops) if (o.kind != 2) return fail(NBODY_ERR_INVALID, "ipc transport: send/recv in a world of one");
            return NBODY_OK;
        }
        // ---- layout of the staged payloads: one slot per send, one per all-gather (every peer reads the same slot)
        struct Out { const char* src; size_t bytes, off; int peer; };   // peer -1: everybody
        std::vector<Out> outs;
        size_t total = 0;
        for (const Op& o : ops) {
            if (o.kind == 1) continue;
            const char* src = o.kind == 2 ? static_cast<const char*>(o.p) + size_t(rank_) * o.bytes : static_cast<const char*>(o.p);
            outs.push_back(Out{src, o.bytes, total, o.kind == 2 ? -1 : o.peer});
            total += (o.bytes + 255) / 256 * 256;
        }
        // the previous contents of the window: who still has to take them?  (asked BEFORE the counters move on)
        FlagList pre{};
        for (int r = 0; r < world_; ++r)
            if (r != rank_ && sent_[r] > 0 && int(flag_load(c_->done[rank_][r]) - (unsigned int)sent_[r]) < 0) {
                pre.flag[pre.n] = &dc_->done[rank_][r].v;
                pre.value[pre.n++] = (unsigned int)sent_[r];
            }
        rc = ensure_window(total);
        if (rc) return rc;
        // ---- host: descriptors of everything that leaves
        bool sends_to[kMaxRanks] = {};
        for (const Out& o : outs)
            for (int r = 0; r < world_; ++r) {
                if (r == rank_ || (o.peer >= 0 && o.peer != r)) continue;
                const uint64_t seq = ++sent_[r];
                if (!host_wait([&] { return seq - c_->consumed[rank_][r].load(std::memory_order_acquire) <= uint64_t(kRing); }))
                    return fail(NBODY_ERR_COMM, "ipc transport: rank " + std::to_string(r) + " stopped reading the messages of rank " + std::to_string(rank_));
                c_->ring[rank_][r][seq % kRing] = Desc{seq, o.bytes, o.off, win_gen_};
                c_->posted[rank_][r].store(seq, std::memory_order_release);
                sends_to[r] = true;
            }
        // ---- stream: window free -> stage -> ready
        if (pre.n) hipLaunchKernelGGL(k_ipc_wait, dim3(1), dim3(64), 0, s, pre, &dc_->abort_word.v, &dc_->dev_timeout[rank_].v, ticks_);
        {
            CopyList cl{};
            size_t biggest = 0;
            auto flush = [&] {
                if (cl.n) hipLaunchKernelGGL(k_ipc_copy, dim3(copy_blocks(biggest), cl.n), dim3(256), 0, s, cl);
                cl.n = 0; biggest = 0;
            };
            for (const Out& o : outs) {
                if (!o.bytes) continue;
                cl.c[cl.n].dst = win_ + o.off; cl.c[cl.n].src = o.src; cl.c[cl.n].bytes = o.bytes;
                biggest = std::max(biggest, o.bytes);
                if (++cl.n == kMaxCopies) flush();
            }
            flush();
        }
        FlagList sig{};
        for (int r = 0; r < world_; ++r)
            if (sends_to[r]) { sig.flag[sig.n] = &dc_->ready[rank_][r].v; sig.value[sig.n++] = (unsigned int)sent_[r]; }
        if (sig.n) hipLaunchKernelGGL(k_ipc_signal, dim3(1), dim3(64), 0, s, sig);
        // ---- host: descriptors of everything that arrives, in the order the operations were issued
        struct In { char* dst; const char* src; size_t bytes; };
        std::vector<In> ins;
        bool recvs_from[kMaxRanks] = {};
        for (const Op& o : ops) {
            if (o.kind == 0) continue;
            for (int r = 0; r < world_; ++r) {
                if (r == rank_ || (o.kind == 1 && o.peer != r)) continue;
                const uint64_t seq = ++recvd_[r];
                if (!host_wait([&] { return c_->posted[r][rank_].load(std::memory_order_acquire) >= seq; }))
                    return fail(NBODY_ERR_COMM, "ipc transport: rank " + std::to_string(rank_) + " expects message " + std::to_string(seq) + " from rank " + std::to_string(r) +
                                                    ", which that rank has not issued within " + std::to_string(int(timeout_s_)) + " s (mismatched exchange?)");
                const Desc d = c_->ring[r][rank_][seq % kRing];
                c_->consumed[r][rank_].store(seq, std::memory_order_release);
                if (d.seq != seq || d.bytes != o.bytes)
                    return fail(NBODY_ERR_COMM, "ipc transport: message " + std::to_string(seq) + " from rank " + std::to_string(r) + " to rank " + std::to_string(rank_) + ": sender has " +
                                                    std::to_string(d.bytes) + " bytes, receiver expects " + std::to_string(o.bytes));
                char* w = nullptr;
                rc = map_window(r, d.gen, &w);
                if (rc) return rc;
                char* dst = o.kind == 2 ? static_cast<char*>(o.p) + size_t(r) * o.bytes : static_cast<char*>(o.p);
                ins.push_back(In{dst, w + d.offset, o.bytes});
                recvs_from[r] = true;
            }
        }
        // ---- stream: ready -> copy out -> done
        FlagList rdy{}, dn{};
        for (int r = 0; r < world_; ++r)
            if (recvs_from[r]) {
                rdy.flag[rdy.n] = &dc_->ready[r][rank_].v; rdy.value[rdy.n++] = (unsigned int)recvd_[r];
                dn.flag[dn.n] = &dc_->done[r][rank_].v; dn.value[dn.n++] = (unsigned int)recvd_[r];
            }
        if (rdy.n) hipLaunchKernelGGL(k_ipc_wait, dim3(1), dim3(64), 0, s, rdy, &dc_->abort_word.v, &dc_->dev_timeout[rank_].v, ticks_);
        {
            CopyList cl{};
            size_t biggest = 0;
            auto flush = [&] {
                if (cl.n) hipLaunchKernelGGL(k_ipc_copy, dim3(copy_blocks(biggest), cl.n), dim3(256), 0, s, cl);
                cl.n = 0; biggest = 0;
            };
            for (const In& i : ins) {
                if (!i.bytes) continue;
                cl.c[cl.n].dst = i.dst; cl.c[cl.n].src = i.src; cl.c[cl.n].bytes = i.bytes;
                biggest = std::max(biggest, i.bytes);
                if (++cl.n == kMaxCopies) flush();
            }
            flush();
        }
        if (dn.n) hipLaunchKernelGGL(k_ipc_signal, dim3(1), dim3(64), 0, s, dn);
        if (hipGetLastError() != hipSuccess) return fail(NBODY_ERR_HIP, "ipc transport: kernel launch failed");
        return NBODY_OK;
    }

    int rank_, world_, device_;
    Ctrl* c_ = nullptr;    // host mapping of the control block
    Ctrl* dc_ = nullptr;   // the same block as the device sees it
    bool joined_ = false, registered_ = false;
    double timeout_s_ = 30.0;
    long long ticks_ = 0;
    int depth_ = 0;
    std::vector<Op> ops_;
    hipStream_t stream_ = nullptr;
    char* win_ = nullptr;
    size_t win_bytes_ = 0;
    uint64_t win_gen_ = 0;
    uint64_t sent_[kMaxRanks] = {}, recvd_[kMaxRanks] = {};
    char* peer_ptr_[kMaxRanks] = {};
    uint64_t peer_gen_[kMaxRanks] = {};
};

}  // namespace

int transport_make_id_ipc(void* id128, std::string* err) {
    static std::atomic<unsigned> counter{0};
    char name[96];
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    std::snprintf(name, sizeof name, "/nbody_ipc_%d_%u_%llx", int(getpid()), counter.fetch_add(1), (unsigned long long)now & 0xffffffffffull);
    int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, off_t(ctrl_bytes())) != 0) {
        if (fd >= 0) { ::close(fd); (void)shm_unlink(name); }
        if (err) *err = "ipc transport: cannot create the shared control block in /dev/shm";
        return NBODY_ERR_COMM;
    }
    void* p = mmap(nullptr, ctrl_bytes(), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    ::close(fd);
    if (p == MAP_FAILED) { (void)shm_unlink(name); if (err) *err = "ipc transport: mmap failed"; return NBODY_ERR_COMM; }
    static_cast<Ctrl*>(p)->magic = kMagic;   // (the rest is the kernel's zero fill)
    (void)munmap(p, ctrl_bytes());
    std::memset(id128, 0, NBODY_COMM_ID_BYTES);
    std::memcpy(id128, kIdTag, sizeof kIdTag);
    std::memcpy(static_cast<char*>(id128) + sizeof kIdTag, name, std::strlen(name) + 1);
    return NBODY_OK;
}

int transport_id_kind(const void* id128) { return std::memcmp(id128, kIdTag, sizeof kIdTag) == 0 ? kTransportIpc : kTransportRccl; }

Transport* transport_create_ipc(const void* id128, int rank, int world, int device, std::string* err, int* code) {
    if (world > kMaxRanks) { if (err) *err = "ipc transport: at most 16 ranks"; if (code) *code = NBODY_ERR_INVALID; return nullptr; }
    char name[NBODY_COMM_ID_BYTES];
    std::memcpy(name, static_cast<const char*>(id128) + sizeof kIdTag, NBODY_COMM_ID_BYTES - sizeof kIdTag);
    name[NBODY_COMM_ID_BYTES - sizeof kIdTag - 1] = 0;
    IpcTransport* t = new IpcTransport(rank, world, device);
    const int rc = t->open(name);
    if (rc) {
        if (err) *err = t->error();
        if (code) *code = rc;
        delete t;
        return nullptr;
    }
    return t;
}

int transport_make_id(int kind, void* id128, std::string* err) {
    return kind == kTransportIpc ? transport_make_id_ipc(id128, err) : transport_make_id_rccl(id128, err);
}

Transport* transport_create(const void* id128, int rank, int world, int device, std::string* err, int* code) {
    return transport_id_kind(id128) == kTransportIpc ? transport_create_ipc(id128, rank, world, device, err, code)
                                                     : transport_create_rccl(id128, rank, world, err, code);
}

}  // namespace nbody
