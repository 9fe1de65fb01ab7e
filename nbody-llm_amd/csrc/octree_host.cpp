// octree_host.cpp -- host octree build + linearisation for the Barnes-Hut path.
//
// Builds the same tree as BarnesHutSimulation::build_tree (src/manual/barnes_hut.rs:143-183),
// bit for bit, but as a stable 8-way radix partition over two ping-pong item arrays instead of a
// Vec per orthant and a Box per node:
//   - 0 bodies -> empty node, 1 body -> leaf {com = pos, mass = m}                    (:145-152)
//   - else classify every body with Bounds::get_orthant (bit i set iff p[i] > center[i],
//     shared.rs:245-254), stable-scatter into the 8 orthant ranges (the reference pushes onto
//     orthants[o] in slice order, :154-158), recurse into non-empty orthants with
//     Bounds::create_orthant (child width = w/2, half = hw/2, centre +- child half,
//     shared.rs:256-272)
//   - mass = sum m, com = sum(pos*m) / mass, both folded left to right over the node's bodies in
//     slice order (:174-179).  Stable partitions keep every range in ascending body id, so the
//     fold order equals the reference's; the sums ride along the classification pass.
// Nodes are emitted in depth-first pre-order with children in orthant order (the order
// calc_force visits them); `skip` = first node after the subtree.
//
// Parallelism: the top `kTaskDepth` levels are partitioned on the calling thread, every subtree
// below becomes a task for the worker pool (the reference forks a rayon task per orthant at
// every level, :160-170), and the per-task node runs are spliced in pre-order afterwards.
//
// Compiled with -ffp-contract=off (x*m must round before it is added).
#include "octree_host.h"

#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#endif
#include "../../include/nbody_hip.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <array>

namespace nbody {

// ------------------------------------------------------------------------------- worker pool
// Task ids are claimed from one monotonically growing counter; a run publishes [limit - n, limit)
// and returns when `done` reaches n.  Workers need not all take part in a run (a sleeping worker
// that wakes late simply finds nothing to claim), so a run costs no wake-up latency when the
// workers are still spinning from the previous one -- a Barnes-Hut step calls run() a few dozen
// times within ~1 ms -- and a late sleeper never delays the caller.
// CPUs of the NUMA node the calling thread is running on ("0-63,128-191" in
// /sys/devices/system/node/nodeN/cpulist); empty if the layout cannot be read
static std::vector<int> numa_node_cpus_of_caller() {
    std::vector<int> cpus;
#if defined(__linux__)
    const int cpu = sched_getcpu();
    if (cpu < 0) return cpus;
    for (int node = 0; node < 64; ++node) {
        char path[96];
        std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
        FILE* f = std::fopen(path, "r");
        if (!f) break;
        char buf[1024] = {0};
        const bool ok = std::fgets(buf, sizeof buf, f) != nullptr;
        std::fclose(f);
        if (!ok) continue;
        std::vector<int> list;
        bool mine = false;
        for (char* p = buf; *p;) {
            char* e = nullptr;
            const long a = std::strtol(p, &e, 10);
            if (e == p) break;
            long b = a;
            if (*e == '-') { p = e + 1; b = std::strtol(p, &e, 10); }
            for (long c = a; c <= b; ++c) { list.push_back(int(c)); if (c == cpu) mine = true; }
            p = (*e == ',') ? e + 1 : e;
            if (*e != ',') break;
        }
        if (mine) return list;
    }
#endif
    return cpus;
}

WorkerPool::WorkerPool(int threads) {
    int extra = std::max(0, threads - 1);
    for (int i = 0; i < extra; ++i) workers_.emplace_back([this] { loop(); });
#if defined(__linux__)
    // keep the workers on the caller's socket: the build is a few hundred microseconds of fork/join
    // over arrays the caller touches too, and a worker across the inter-socket link doubles its
    // memory latency (step-to-step build times of 0.8-1.2 ms on a 2-socket host).  A mask, not one
    // core per thread: the scheduler keeps its freedom inside the node.  NBODY_POOL_NUMA=0 disables it.
    const char* env = std::getenv("NBODY_POOL_NUMA");
    if (extra > 0 && !(env && env[0] == '0')) {
        const std::vector<int> cpus = numa_node_cpus_of_caller();
        if (int(cpus.size()) >= threads) {
            cpu_set_t set;
            CPU_ZERO(&set);
            for (int c : cpus) if (c < CPU_SETSIZE) CPU_SET(c, &set);
            for (auto& w : workers_) (void)pthread_setaffinity_np(w.native_handle(), sizeof set, &set);
        }
    }
#endif
}

WorkerPool::~WorkerPool() {
    {
        std::lock_guard<std::mutex> lk(m_);
        stop_.store(true);
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
}

bool WorkerPool::try_one() {
    long long t = next_.load(std::memory_order_acquire);
    for (;;) {
        const long long lim = limit_.load(std::memory_order_acquire);
        if (t >= lim) return false;
        if (next_.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel)) {
            // the run that owns id t is still open (its caller waits for done_), so fn_/n_ are its own
            const std::function<void(int)>* fn = fn_.load(std::memory_order_acquire);
            (*fn)(int(t - (lim - n_.load(std::memory_order_acquire))));
            done_.fetch_add(1, std::memory_order_acq_rel);
            return true;
        }
    }
}

bool WorkerPool::try_background() {
    int expect = 1;
    if (bg_state_.load(std::memory_order_acquire) != 1 ||
        !bg_state_.compare_exchange_strong(expect, 2, std::memory_order_acq_rel)) return false;
    (*bg_fn_.load(std::memory_order_acquire))();
    bg_state_.store(3, std::memory_order_release);
    return true;
}

void WorkerPool::post_background(const std::function<void()>& fn) {
    bg_fn_.store(&fn, std::memory_order_release);
    bool wake;
    {
        std::lock_guard<std::mutex> lk(m_);
        bg_state_.store(1, std::memory_order_release);
        wake = sleepers_ > 0;
    }
    if (wake) cv_.notify_one();
}

void WorkerPool::wait_background() {
    if (bg_state_.load(std::memory_order_acquire) == 0) return;
    try_background();  // nobody picked it up yet (no workers, or all busy): do it here
    while (bg_state_.load(std::memory_order_acquire) != 3) __builtin_ia32_pause();
    bg_state_.store(0, std::memory_order_release);
}

void WorkerPool::loop() {
    for (;;) {
        int idle = 0;
        while (!stop_.load(std::memory_order_relaxed)) {
            if (try_background()) { idle = 0; continue; }
            if (try_one()) { idle = 0; continue; }
            if (++idle > 2000) break;  // ~50 us without work: go to sleep
            __builtin_ia32_pause();
        }
        if (stop_.load(std::memory_order_relaxed)) return;
        std::unique_lock<std::mutex> lk(m_);
        sleepers_++;
        cv_.wait(lk, [&] { return stop_.load() || next_.load() < limit_.load() || bg_state_.load() == 1; });
        sleepers_--;
        if (stop_.load()) return;
    }
}

void WorkerPool::run(int n_tasks, const std::function<void(int)>& fn) {
    if (n_tasks <= 0) return;
    if (workers_.empty() || n_tasks == 1) {
        for (int t = 0; t < n_tasks; ++t) fn(t);
        return;
    }
    done_.store(0, std::memory_order_relaxed);
    fn_.store(&fn, std::memory_order_release);
    n_.store(n_tasks, std::memory_order_release);
    bool wake;
    {
        std::lock_guard<std::mutex> lk(m_);  // pairs with the sleepers' predicate check
        limit_.fetch_add(n_tasks, std::memory_order_acq_rel);
        wake = sleepers_ > 0;
    }
    if (wake) cv_.notify_all();
    while (try_one()) {}
    while (done_.load(std::memory_order_acquire) != n_tasks) __builtin_ia32_pause();
}

// ------------------------------------------------------------------------------ output arrays
template <class T>
void HostTreeT<T>::clear() {
    auto rel = release ? release : +[](void* p) { std::free(p); };
    if (nodes) rel(nodes);
    if (order) rel(order);
    nodes = nullptr; order = nullptr;
    cap_nodes = cap_order = n_nodes = n_order = 0;
}

template <class T>
void HostTreeT<T>::reserve(size_t n_nodes_wanted, size_t order_n) {
    auto al = alloc ? alloc : +[](size_t n) { return std::malloc(n); };
    auto rel = release ? release : +[](void* p) { std::free(p); };
    if (n_nodes_wanted > cap_nodes) {
        size_t cap = n_nodes_wanted + n_nodes_wanted / 4 + 64;
        if (nodes) rel(nodes);
        nodes = static_cast<NodeRecT<T>*>(al(cap * sizeof(NodeRecT<T>)));
        cap_nodes = cap;
    }
    if (order_n > cap_order) {
        size_t cap = order_n + order_n / 4 + 64;
        if (order) rel(order);
        order = static_cast<int32_t*>(al(cap * sizeof(int32_t)));
        cap_order = cap;
    }
}

// --------------------------------------------------------------------------------- the build
namespace {

template <class T> struct ItemT { T x, y, z, m; int32_t id; };

template <class T>
struct BoxT {
    T c[3];
    T hw, w;
    BoxT child(int o) const {  // Bounds::create_orthant
        BoxT b;
        b.w = w * T(0.5);
        b.hw = hw * T(0.5);
        for (int i = 0; i < 3; ++i) b.c[i] = (o >> i & 1) ? c[i] + b.hw : c[i] - b.hw;
        return b;
    }
};

template <class T>
struct EmitT {
    std::vector<NodeRecT<T>> nodes;
    std::vector<int32_t> order;
    bool too_deep = false;
    int max_depth = 0;
};

// classify + fold; returns per-orthant counts in cnt[8], the node's mass and com
template <class T>
inline void classify_and_sum(const ItemT<T>* src, uint8_t* code, int n, const BoxT<T>& box, int cnt[8], NodeAT<T>& node) {
    for (int o = 0; o < 8; ++o) cnt[o] = 0;
    T mass = 0, sx = 0, sy = 0, sz = 0;
    const T cx = box.c[0], cy = box.c[1], cz = box.c[2];
    for (int k = 0; k < n; ++k) {
        const ItemT<T>& it = src[k];
        int o = (it.x > cx ? 1 : 0) | (it.y > cy ? 2 : 0) | (it.z > cz ? 4 : 0);
        code[k] = uint8_t(o);
        cnt[o]++;
        mass += it.m;
        sx += it.x * it.m;
        sy += it.y * it.m;
        sz += it.z * it.m;
    }
    node.m = mass;
    node.x = sx / mass;
    node.y = sy / mass;
    node.z = sz / mass;
}

template <class T>
inline void scatter(const ItemT<T>* src, ItemT<T>* dst, const uint8_t* code, int n, const int cnt[8], int start[8]) {
    int off[8];
    int run = 0;
    for (int o = 0; o < 8; ++o) { start[o] = off[o] = run; run += cnt[o]; }
    for (int k = 0; k < n; ++k) dst[off[code[k]]++] = src[k];
}

// pn, gpn: bodies in the parent and grandparent cells (NodeB::hot)
template <class T>
void build_rec(ItemT<T>* src, ItemT<T>* tmp, uint8_t* code, int n, const BoxT<T>& box, int depth, EmitT<T>& e, int pn, int gpn) {
    const int me = int(e.nodes.size());
    if (depth > e.max_depth) e.max_depth = depth;
    e.nodes.push_back(NodeRecT<T>{NodeAT<T>{0, 0, 0, 0}, NodeBT<T>{box.w * box.w, me + 1, gpn, -1}});
    if (n == 0) return;
    if (n == 1) {
        e.nodes[me].a = NodeAT<T>{src[0].x, src[0].y, src[0].z, src[0].m};
        e.nodes[me].b.body = src[0].id;
        e.order.push_back(src[0].id);
        return;
    }
    if (depth >= NBODY_MAX_TREE_DEPTH) { e.too_deep = true; return; }
    int cnt[8], start[8];
    NodeAT<T> node;
    classify_and_sum(src, code, n, box, cnt, node);
    scatter(src, tmp, code, n, cnt, start);
    for (int o = 0; o < 8; ++o)
        if (cnt[o]) build_rec(tmp + start[o], src + start[o], code + start[o], cnt[o], box.child(o), depth + 1, e, n, pn);
    e.nodes[me].a = node;
    e.nodes[me].b.skip = int(e.nodes.size());
}

constexpr int kTaskDepth = 2;

template <class T> struct TaskT { ItemT<T>* src; ItemT<T>* tmp; uint8_t* code; int n; BoxT<T> box; int depth; EmitT<T>* out; int pn, gpn; };

template <class T>
struct TopEntryT {
    int task = -1;   // >= 0: the subtree built by that task; else a node of the top levels
    NodeAT<T> a{};
    NodeBT<T> b{};
    int end = 0;     // node entries: index of the first entry after this node's subtree
    int depth = 0;
};

template <class T>
void build_top(ItemT<T>* src, ItemT<T>* tmp, uint8_t* code, int n, const BoxT<T>& box, int depth, std::vector<TopEntryT<T>>& top,
               std::vector<TaskT<T>>& tasks, int pn, int gpn) {
    const int me = int(top.size());
    top.emplace_back();
    top[me].b = NodeBT<T>{box.w * box.w, 0, gpn, -1};
    top[me].depth = depth;
    if (n == 0) { top[me].end = me + 1; return; }
    if (n == 1) {
        top[me].a = NodeAT<T>{src[0].x, src[0].y, src[0].z, src[0].m};
        top[me].b.body = src[0].id;
        top[me].end = me + 1;
        return;
    }
    if (depth >= kTaskDepth) {
        top[me].task = int(tasks.size());
        top[me].end = me + 1;
        tasks.push_back(TaskT<T>{src, tmp, code, n, box, depth, nullptr, pn, gpn});
        return;
    }
    int cnt[8], start[8];
    NodeAT<T> node;
    classify_and_sum(src, code, n, box, cnt, node);
    scatter(src, tmp, code, n, cnt, start);
    for (int o = 0; o < 8; ++o)
        if (cnt[o]) build_top(tmp + start[o], src + start[o], code + start[o], cnt[o], box.child(o), depth + 1, top, tasks, n, pn);
    top[me].a = node;
    top[me].end = int(top.size());
}

}  // namespace

template <class T>
struct BuildScratchT<T>::Impl {
    std::vector<ItemT<T>> buf_a, buf_b;
    std::vector<uint8_t> code;
    std::vector<EmitT<T>> emits;  // one per subtree task; capacities survive from step to step
};
template <class T> BuildScratchT<T>::BuildScratchT() : impl(new Impl) {}
template <class T> BuildScratchT<T>::~BuildScratchT() { delete impl; }

template <class T>
void build_octree(const T* pos4, int n_seg, int seg_cap, const int* count, const T center[3], T width,
                  WorkerPool& pool, BuildScratchT<T>& scratch, HostTreeT<T>& out) {
    using Item = ItemT<T>;
    using Box = BoxT<T>;
    using Emit = EmitT<T>;
    using Task = TaskT<T>;
    using TopEntry = TopEntryT<T>;
    using NodeA = NodeAT<T>;
    using NodeB = NodeBT<T>;
    using NodeRec = NodeRecT<T>;
    size_t n = 0;
    std::vector<size_t> seg_first(n_seg + 1, 0);
    for (int s = 0; s < n_seg; ++s) { seg_first[s] = n; n += size_t(count[s]); }
    seg_first[n_seg] = n;
    // all working memory is kept between calls: in a VM a fresh 10 MB costs more in page faults
    // than the build itself
    std::vector<Item>& buf_a = scratch.impl->buf_a;
    std::vector<Item>& buf_b = scratch.impl->buf_b;
    std::vector<uint8_t>& code = scratch.impl->code;
    std::vector<Emit>& emits = scratch.impl->emits;
    if (buf_a.size() < n) { buf_a.resize(n); buf_b.resize(n); code.resize(n); }
    Item* A = buf_a.data();
    Item* B = buf_b.data();
    uint8_t* C = code.data();

    Box root;
    root.c[0] = center[0]; root.c[1] = center[1]; root.c[2] = center[2];
    root.hw = width * T(0.5);  // Bounds::new, shared.rs:236-243
    root.w = width;

    const int n_thr = pool.size();
    static const bool timing = std::getenv("NBODY_TREE_TIMING") != nullptr;
    auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "  octree %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_start).count());
        t_start = now;
    };
    auto fill = [&](size_t k0, size_t k1) {  // bodies enter in id order = the reference's vector order
        int s = 0;
        while (seg_first[s + 1] <= k0 && s + 1 < n_seg) ++s;
        for (size_t k = k0; k < k1; ++k) {
            while (k >= seg_first[s + 1]) ++s;
            const size_t j = k - seg_first[s];
            const T* p = pos4 + 4 * (size_t(s) * seg_cap + j);
            A[k] = Item{p[0], p[1], p[2], p[3], int32_t(size_t(s) * seg_cap + j)};
        }
    };

    std::vector<TopEntry> top;
    std::vector<Task> tasks;
    if (n < 8192 || n_thr == 1) {
        fill(0, n);
        build_top(A, B, C, int(n), root, 0, top, tasks, int(n), int(n));
    } else {
        // ---- big nodes (more than `big` bodies) are partitioned level by level with every thread
        // working on chunks of them; their folds (mass, com: sequential, the reference's order) run
        // as separate tasks beside the chunk work.  Everything smaller becomes a subtree task.
        // Depth alone is a poor cut: a Plummer sphere in a wide box keeps most bodies in 8 cells
        // per level for several levels.
        struct Big { Item* src; Item* tmp; uint8_t* code; int n; Box box; int depth; int self; int pn; int fold_from; };  // pn: bodies in the parent cell
        // fold_from >= 0: the cell holds exactly the bodies of that ancestor, in the same order (a cluster much smaller than
        // the box sits in one child per level for several levels): same sums, same bits -- copied, and nothing was moved
        struct TNode { NodeA a{}; NodeB b{}; int child[8]; int task = -1; int depth = 0; };
        std::vector<TNode> tn;
        auto new_tnode = [&](const Box& bx, int hot, int depth) { TNode t; t.b = NodeB{bx.w * bx.w, 0, hot, -1}; t.depth = depth; for (int& c : t.child) c = -1; tn.push_back(t); return int(tn.size()) - 1; };
        const int big = std::max(2048, int(n / 64));
        const int chunk = std::max(1024, big / 2);
        // the root's fold is a 4 x n-long dependent chain (~50 us at n = 65 536) nobody needs before the
        // flatten below: it runs beside all the levels, over the caller's array (never written here;
        // id order = the order fill() puts the bodies in)
        NodeA root_fold{};
        const std::function<void()> fold_root = [&] {
            T mass = 0, sx = 0, sy = 0, sz = 0;
            for (int sg = 0; sg < n_seg; ++sg) {
                const T* p = pos4 + 4 * (size_t(sg) * seg_cap);
                for (int j = 0; j < count[sg]; ++j, p += 4) { mass += p[3]; sx += p[0] * p[3]; sy += p[1] * p[3]; sz += p[2] * p[3]; }
            }
            root_fold = NodeA{sx / mass, sy / mass, sz / mass, mass};
        };
        pool.post_background(fold_root);
        pool.run(n_thr, [&](int t) { const size_t c = (n + n_thr - 1) / n_thr; fill(std::min(n, size_t(t) * c), std::min(n, size_t(t + 1) * c)); });
        lap("fill");
        std::vector<Big> level{Big{A, B, C, int(n), root, 0, new_tnode(root, int(n), 0), int(n), -1}};
        struct Piece { int node; int k0, k1; };
        while (!level.empty()) {
            std::vector<Piece> pieces;
            for (int b = 0; b < int(level.size()); ++b)
                for (int k0 = 0; k0 < level[b].n; k0 += chunk) pieces.push_back(Piece{b, k0, std::min(level[b].n, k0 + chunk)});
            std::vector<std::array<int, 8>> pcnt(pieces.size());
            const int NP = int(pieces.size()), NB = int(level.size());
            pool.run(NP + NB, [&](int t) {
                if (t < NB) {  // the node's fold (handed out first: they are the long tasks of the run)
                    const Big& g = level[t];
                    if (g.depth == 0) return;  // the root's runs in the background
                    if (g.fold_from >= 0) { tn[g.self].a = tn[g.fold_from].a; return; }
                    T mass = 0, sx = 0, sy = 0, sz = 0;
                    for (int k = 0; k < g.n; ++k) { const Item& it = g.src[k]; mass += it.m; sx += it.x * it.m; sy += it.y * it.m; sz += it.z * it.m; }
                    tn[g.self].a = NodeA{sx / mass, sy / mass, sz / mass, mass};
                    return;
                }
                const Piece& pc = pieces[t - NB];
                const Big& g = level[pc.node];
                std::array<int, 8> c{};
                for (int k = pc.k0; k < pc.k1; ++k) {
                    const Item& it = g.src[k];
                    int o = (it.x > g.box.c[0] ? 1 : 0) | (it.y > g.box.c[1] ? 2 : 0) | (it.z > g.box.c[2] ? 4 : 0);
                    g.code[k] = uint8_t(o);
                    c[o]++;
                }
                pcnt[t - NB] = c;
            });
            // per node: orthant starts, then per piece: its write offsets (stable: pieces in order)
            std::vector<std::array<int, 8>> nstart(NB), ntot(NB), poff(NP);
            for (int b = 0; b < NB; ++b) ntot[b].fill(0);
            for (int t = 0; t < NP; ++t) for (int o = 0; o < 8; ++o) ntot[pieces[t].node][o] += pcnt[t][o];
            for (int b = 0; b < NB; ++b) { int run = 0; for (int o = 0; o < 8; ++o) { nstart[b][o] = run; run += ntot[b][o]; } }
            { std::vector<std::array<int, 8>> run = nstart;
              for (int t = 0; t < NP; ++t) for (int o = 0; o < 8; ++o) { poff[t][o] = run[pieces[t].node][o]; run[pieces[t].node][o] += pcnt[t][o]; } }
            // a node all of whose bodies went to ONE child: nothing to move, the child works on the same arrays
            std::vector<char> same(NB, 0);
            bool any_move = false;
            for (int b = 0; b < NB; ++b) {
                for (int o = 0; o < 8; ++o) if (ntot[b][o] == level[b].n) same[b] = 1;
                if (!same[b]) any_move = true;
            }
            if (any_move) pool.run(NP, [&](int t) {
                const Piece& pc = pieces[t];
                const Big& g = level[pc.node];
                if (same[pc.node]) return;
                int o8[8];
                for (int o = 0; o < 8; ++o) o8[o] = poff[t][o];
                for (int k = pc.k0; k < pc.k1; ++k) g.tmp[o8[g.code[k]]++] = g.src[k];
            });
            std::vector<Big> next;
            for (int b = 0; b < NB; ++b) {
                const Big g = level[b];
                for (int o = 0; o < 8; ++o) {
                    const int cn = ntot[b][o];
                    if (!cn) continue;
                    const Box cb = g.box.child(o);
                    const int id = new_tnode(cb, g.pn, g.depth + 1);  // the child's grandparent is g's parent
                    tn[g.self].child[o] = id;
                    Item* csrc = (same[b] ? g.src : g.tmp) + nstart[b][o];
                    Item* ctmp = (same[b] ? g.tmp : g.src) + nstart[b][o];
                    uint8_t* ccode = g.code + nstart[b][o];
                    if (cn == 1) {
                        tn[id].a = NodeA{csrc[0].x, csrc[0].y, csrc[0].z, csrc[0].m};
                        tn[id].b.body = csrc[0].id;
                    } else if (cn > big && g.depth + 1 < 24) {
                        // (the root's own fold runs in the background and lands after the loop: its only child folds itself)
                        const int from = (same[b] && g.depth > 0) ? (g.fold_from >= 0 ? g.fold_from : g.self) : -1;
                        next.push_back(Big{csrc, ctmp, ccode, cn, cb, g.depth + 1, id, g.n, from});
                    } else {
                        tn[id].task = int(tasks.size());
                        tasks.push_back(Task{csrc, ctmp, ccode, cn, cb, g.depth + 1, nullptr, g.n, g.pn});
                    }
                }
            }
            level.swap(next);
            if (timing) { char nm[32]; std::snprintf(nm, sizeof nm, "L nb=%d np=%d", NB, NP); lap(nm); }
        }
        pool.wait_background();
        tn[0].a = root_fold;
        // flatten the big-node tree in pre-order
        std::vector<int> stack{0};
        std::vector<int> open_end;  // entries whose `end` is patched when their subtree is complete
        std::function<void(int)> emit = [&](int id) {
            const int me = int(top.size());
            top.emplace_back();
            top[me].a = tn[id].a;
            top[me].b = tn[id].b;
            top[me].task = tn[id].task;
            top[me].depth = tn[id].depth;
            for (int o = 0; o < 8; ++o) if (tn[id].child[o] >= 0) emit(tn[id].child[o]);
            top[me].end = int(top.size());
        };
        emit(0);
    }

    lap("top");
    // larger subtrees first: the dynamic hand-out then ends on small ones
    std::vector<int> by_size(tasks.size());
    for (size_t i = 0; i < tasks.size(); ++i) by_size[i] = int(i);
    std::sort(by_size.begin(), by_size.end(), [&](int x, int y) { return tasks[x].n > tasks[y].n; });
    if (emits.size() < tasks.size()) emits.resize(tasks.size());
    for (size_t i = 0; i < tasks.size(); ++i) tasks[i].out = &emits[i];
    pool.run(int(tasks.size()), [&](int i) {
        Task& tk = tasks[by_size[i]];
        Emit& e = *tk.out;
        e.nodes.clear(); e.order.clear(); e.too_deep = false; e.max_depth = tk.depth;
        build_rec(tk.src, tk.tmp, tk.code, tk.n, tk.box, tk.depth, e, tk.pn, tk.gpn);
    });

    lap("subtrees");
    // splice: final index of every top entry = prefix sum of entry sizes
    std::vector<int> first(top.size() + 1, 0);
    std::vector<int> ofirst(top.size() + 1, 0);
    for (size_t e = 0; e < top.size(); ++e) {
        int sz = top[e].task >= 0 ? int(tasks[top[e].task].out->nodes.size()) : 1;
        int osz = top[e].task >= 0 ? int(tasks[top[e].task].out->order.size()) : (top[e].b.body >= 0 ? 1 : 0);
        first[e + 1] = first[e] + sz;
        ofirst[e + 1] = ofirst[e] + osz;
    }
    const int total = first[top.size()];
    out.reserve(size_t(total), size_t(ofirst[top.size()]));
    out.n_nodes = size_t(total);
    out.n_order = size_t(ofirst[top.size()]);
    out.too_deep = false;
    for (size_t e = 0; e < top.size(); ++e) {
        if (top[e].task < 0) {
            NodeB b = top[e].b;
            b.skip = first[top[e].end];
            out.nodes[first[e]] = NodeRec{top[e].a, b};
            if (b.body >= 0) out.order[ofirst[e]] = b.body;
        }
    }
    pool.run(int(top.size()), [&](int e) {
        if (top[e].task < 0) return;
        const Emit& em = *tasks[top[e].task].out;
        const int base = first[e];
        for (size_t i = 0; i < em.nodes.size(); ++i) {
            NodeRec r = em.nodes[i];
            r.b.skip += base;
            out.nodes[base + i] = r;
        }
        if (!em.order.empty()) std::memcpy(&out.order[ofirst[e]], em.order.data(), em.order.size() * sizeof(int32_t));
    });
    out.max_depth = 0;
    for (auto& tk : tasks) { out.too_deep = out.too_deep || tk.out->too_deep; out.max_depth = std::max(out.max_depth, tk.out->max_depth); }
    for (const TopEntry& te : top) if (te.task < 0) out.max_depth = std::max(out.max_depth, te.depth);   // nodes above the subtree tasks
    lap("splice");
}

template struct HostTreeT<float>;
template struct HostTreeT<double>;
template struct BuildScratchT<float>;
template struct BuildScratchT<double>;
template void build_octree<float>(const float*, int, int, const int*, const float[3], float, WorkerPool&, BuildScratchT<float>&, HostTreeT<float>&);
template void build_octree<double>(const double*, int, int, const int*, const double[3], double, WorkerPool&, BuildScratchT<double>&, HostTreeT<double>&);

}  // namespace nbody
