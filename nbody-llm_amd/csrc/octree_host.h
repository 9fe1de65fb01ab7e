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
    // one background task beside the runs: taken by the first worker that looks for work (or by the
    // caller in wait_background if none did); fn must stay alive until wait_background returns
    void post_background(const std::function<void()>& fn);
    void wait_background();

private:
    void loop();
    bool try_one();
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<const std::function<void(int)>*> fn_{nullptr};
    std::atomic<long long> next_{0};    // next task id to claim (grows over the pool's lifetime)
    std::atomic<long long> limit_{0};   // ids below this are published
    std::atomic<int> n_{0};             // size of the open run
    std::atomic<int> done_{0};          // tasks of the open run that have finished
    std::atomic<bool> stop_{false};
    std::atomic<const std::function<void()>*> bg_fn_{nullptr};
    std::atomic<int> bg_state_{0};      // 0 none, 1 posted, 2 running, 3 done
    bool try_background();
    int sleepers_ = 0;                  // under m_
};

// One record per node, templated on the reference's F (f32 or f64; shared.rs:12-44): a single 32-byte (f32) or
// 64-byte (f64) load on the device.
template <class T> struct NodeAT { T x, y, z, m; };   // centre of mass, mass
// width^2, skip, "hot" = bodies in the node's GRANDPARENT cell (the root's count for depths 0 and 1), body id of
// a leaf or -1.  The width itself is sqrt(w2), exactly (IEEE sqrt of a rounded square returns the operand).
// hot ranks the nodes by how many walks visit them (a node is visited by the bodies that open its parent,
// i.e. those within parent-width / theta of it: about the population around the grandparent cell); the
// LDS-staged walk keeps the highest-ranked records in LDS (kernels_bh.hip, variant 3).  tools/bh_visit_hist.py: the
// 2 048 nodes ranked first by this score take 66 % of all visits at N = 65 536, the 2 048 truly hottest 67 %.
template <class T> struct NodeBT { T w2; int32_t skip; int32_t hot; int32_t body; };
template <class T> struct alignas(sizeof(T) * 8) NodeRecT { NodeAT<T> a; NodeBT<T> b; };
using NodeA = NodeAT<float>;
using NodeB = NodeBT<float>;
using NodeRec = NodeRecT<float>;
static_assert(sizeof(NodeRecT<float>) == 32 && sizeof(NodeRecT<double>) == 64, "node record layout");

// Output arrays live in caller-chosen memory (the API hands in pinned-host allocators so the
// H2D copy of the node array is a single DMA; tests use malloc).
template <class T>
struct HostTreeT {
    NodeRecT<T>* nodes = nullptr;
    int32_t* order = nullptr;  // body ids in depth-first leaf order
    size_t n_nodes = 0, n_order = 0;
    size_t cap_nodes = 0, cap_order = 0;
    bool too_deep = false;
    int max_depth = 0;         // depth of the deepest node (root = 0)
    void* (*alloc)(size_t) = nullptr;  // null -> malloc/free
    void (*release)(void*) = nullptr;
    void reserve(size_t n_nodes_wanted, size_t order_n);
    void clear();
    ~HostTreeT() { clear(); }
    HostTreeT() = default;
    HostTreeT(const HostTreeT&) = delete;
    HostTreeT& operator=(const HostTreeT&) = delete;
};
using HostTree = HostTreeT<float>;

// pos4: {x,y,z,m} records; the bodies are the concatenation of n_seg segments of seg_cap slots
// holding count[s] live bodies each; a body's id is s*seg_cap + j.  Bodies enter the build in id
// order, which is the reference's vector order.
// working memory of the build, kept from step to step
template <class T>
struct BuildScratchT {
    struct Impl;
    Impl* impl;
    BuildScratchT();
    ~BuildScratchT();
    BuildScratchT(const BuildScratchT&) = delete;
    BuildScratchT& operator=(const BuildScratchT&) = delete;
};
using BuildScratch = BuildScratchT<float>;

template <class T>
void build_octree(const T* pos4, int n_seg, int seg_cap, const int* count, const T center[3], T width,
                  WorkerPool& pool, BuildScratchT<T>& scratch, HostTreeT<T>& out);

}  // namespace nbody
