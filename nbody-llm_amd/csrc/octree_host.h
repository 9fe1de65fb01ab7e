// octree_host.h -- host octree build + linearisation (BarnesHutSimulation::build_tree,
// src/manual/barnes_hut.rs:143-183), internal to libnbody_hip.so.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <functional>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>

namespace nbody {

// Persistent worker pool: the reference sizes a global rayon pool once from `-t`
// (src/main.rs:46-50); spawning threads per step would cost more than a small tree build.
class WorkerPool {
public:
    explicit WorkerPool(int threads);
    ~WorkerPool();
    int size() const { return int(workers_.size()) + 1; }  // workers + the calling thread
    // runs fn(task) for task in [0, n_tasks), tasks handed out dynamically; returns when all done
    void run(int n_tasks, const std::function<void(int)>& fn);

private:
    void loop();
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int)>* fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_tasks_ = 0;
    int active_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

struct NodeA { float x, y, z, m; };               // centre of mass, mass
struct NodeB { float w2; int32_t skip; float w; int32_t body; };  // width^2, skip, width, body id of a leaf or -1

// Output arrays live in caller-chosen memory (the API hands in pinned-host allocators so the
// H2D copy of the node array is a single DMA; tests use malloc).
struct HostTree {
    NodeA* a = nullptr;
    NodeB* b = nullptr;
    int32_t* order = nullptr;  // body ids in depth-first leaf order
    size_t n_nodes = 0, n_order = 0;
    size_t cap_nodes = 0, cap_order = 0;
    bool too_deep = false;
    void* (*alloc)(size_t) = nullptr;  // null -> malloc/free
    void (*release)(void*) = nullptr;
    void reserve(size_t nodes, size_t order_n);
    void clear();
    ~HostTree() { clear(); }
    HostTree() = default;
    HostTree(const HostTree&) = delete;
    HostTree& operator=(const HostTree&) = delete;
};

// pos4: {x,y,z,m} records; the bodies are the concatenation of n_seg segments of seg_cap slots
// holding count[s] live bodies each; a body's id is s*seg_cap + j.  Bodies enter the build in id
// order, which is the reference's vector order.
void build_octree(const float* pos4, int n_seg, int seg_cap, const int* count, const float center[3], float width,
                  WorkerPool& pool, HostTree& out);

}  // namespace nbody
