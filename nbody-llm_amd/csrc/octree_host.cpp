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
#include "../../include/nbody_hip.h"

#include <cstring>
#include <cstdlib>
#include <algorithm>

namespace nbody {

// ------------------------------------------------------------------------------- worker pool
WorkerPool::WorkerPool(int threads) {
    int extra = std::max(0, threads - 1);
    for (int i = 0; i < extra; ++i) workers_.emplace_back([this] { loop(); });
}

WorkerPool::~WorkerPool() {
    {
        std::lock_guard<std::mutex> lk(m_);
        stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
}

void WorkerPool::loop() {
    uint64_t seen = 0;
    for (;;) {
        const std::function<void(int)>* fn;
        int n;
        {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
            if (stop_) return;
            seen = epoch_;
            fn = fn_;
            n = n_tasks_;
        }
        for (;;) {
            int t = next_.fetch_add(1, std::memory_order_relaxed);
            if (t >= n) break;
            (*fn)(t);
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            if (--active_ == 0) done_cv_.notify_all();
        }
    }
}

void WorkerPool::run(int n_tasks, const std::function<void(int)>& fn) {
    if (n_tasks <= 0) return;
    if (workers_.empty() || n_tasks == 1) {
        for (int t = 0; t < n_tasks; ++t) fn(t);
        return;
    }
    {
        std::lock_guard<std::mutex> lk(m_);
        fn_ = &fn;
        n_tasks_ = n_tasks;
        next_.store(0, std::memory_order_relaxed);
        active_ = int(workers_.size());
        ++epoch_;
    }
    cv_.notify_all();
    for (;;) {  // the caller works too
        int t = next_.fetch_add(1, std::memory_order_relaxed);
        if (t >= n_tasks) break;
        fn(t);
    }
    std::unique_lock<std::mutex> lk(m_);
    done_cv_.wait(lk, [&] { return active_ == 0; });
}

// ------------------------------------------------------------------------------ output arrays
void HostTree::clear() {
    auto rel = release ? release : +[](void* p) { std::free(p); };
    if (a) rel(a);
    if (b) rel(b);
    if (order) rel(order);
    a = nullptr; b = nullptr; order = nullptr;
    cap_nodes = cap_order = n_nodes = n_order = 0;
}

void HostTree::reserve(size_t nodes, size_t order_n) {
    auto al = alloc ? alloc : +[](size_t n) { return std::malloc(n); };
    auto rel = release ? release : +[](void* p) { std::free(p); };
    if (nodes > cap_nodes) {
        size_t cap = nodes + nodes / 4 + 64;
        if (a) rel(a);
        if (b) rel(b);
        a = static_cast<NodeA*>(al(cap * sizeof(NodeA)));
        b = static_cast<NodeB*>(al(cap * sizeof(NodeB)));
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

struct Item { float x, y, z, m; int32_t id; };

struct Box {
    float c[3];
    float hw, w;
    Box child(int o) const {  // Bounds::create_orthant
        Box b;
        b.w = w * 0.5f;
        b.hw = hw * 0.5f;
        for (int i = 0; i < 3; ++i) b.c[i] = (o >> i & 1) ? c[i] + b.hw : c[i] - b.hw;
        return b;
    }
};

struct Emit {
    std::vector<NodeA> a;
    std::vector<NodeB> b;
    std::vector<int32_t> order;
    bool too_deep = false;
};

// classify + fold; returns per-orthant counts in cnt[8], the node's mass and com
inline void classify_and_sum(const Item* src, uint8_t* code, int n, const Box& box, int cnt[8], NodeA& node) {
    for (int o = 0; o < 8; ++o) cnt[o] = 0;
    float mass = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
    const float cx = box.c[0], cy = box.c[1], cz = box.c[2];
    for (int k = 0; k < n; ++k) {
        const Item& it = src[k];
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

inline void scatter(const Item* src, Item* dst, const uint8_t* code, int n, const int cnt[8], int start[8]) {
    int off[8];
    int run = 0;
    for (int o = 0; o < 8; ++o) { start[o] = off[o] = run; run += cnt[o]; }
    for (int k = 0; k < n; ++k) dst[off[code[k]]++] = src[k];
}

void build_rec(Item* src, Item* tmp, uint8_t* code, int n, const Box& box, int depth, Emit& e) {
    const int me = int(e.a.size());
    e.a.push_back(NodeA{0.f, 0.f, 0.f, 0.f});
    e.b.push_back(NodeB{box.w * box.w, me + 1, box.w, -1});
    if (n == 0) return;
    if (n == 1) {
        e.a[me] = NodeA{src[0].x, src[0].y, src[0].z, src[0].m};
        e.b[me].body = src[0].id;
        e.order.push_back(src[0].id);
        return;
    }
    if (depth >= NBODY_MAX_TREE_DEPTH) { e.too_deep = true; return; }
    int cnt[8], start[8];
    NodeA node;
    classify_and_sum(src, code, n, box, cnt, node);
    scatter(src, tmp, code, n, cnt, start);
    for (int o = 0; o < 8; ++o)
        if (cnt[o]) build_rec(tmp + start[o], src + start[o], code + start[o], cnt[o], box.child(o), depth + 1, e);
    e.a[me] = node;
    e.b[me].skip = int(e.a.size());
}

constexpr int kTaskDepth = 2;

struct Task { Item* src; Item* tmp; uint8_t* code; int n; Box box; int depth; Emit out; };

struct TopEntry {
    int task = -1;   // >= 0: the subtree built by that task; else a node of the top levels
    NodeA a{};
    NodeB b{};
    int end = 0;     // node entries: index of the first entry after this node's subtree
};

void build_top(Item* src, Item* tmp, uint8_t* code, int n, const Box& box, int depth, std::vector<TopEntry>& top,
               std::vector<Task>& tasks) {
    const int me = int(top.size());
    top.emplace_back();
    top[me].b = NodeB{box.w * box.w, 0, box.w, -1};
    if (n == 0) { top[me].end = me + 1; return; }
    if (n == 1) {
        top[me].a = NodeA{src[0].x, src[0].y, src[0].z, src[0].m};
        top[me].b.body = src[0].id;
        top[me].end = me + 1;
        return;
    }
    if (depth >= kTaskDepth) {
        top[me].task = int(tasks.size());
        top[me].end = me + 1;
        tasks.push_back(Task{src, tmp, code, n, box, depth, Emit{}});
        return;
    }
    int cnt[8], start[8];
    NodeA node;
    classify_and_sum(src, code, n, box, cnt, node);
    scatter(src, tmp, code, n, cnt, start);
    for (int o = 0; o < 8; ++o)
        if (cnt[o]) build_top(tmp + start[o], src + start[o], code + start[o], cnt[o], box.child(o), depth + 1, top, tasks);
    top[me].a = node;
    top[me].end = int(top.size());
}

}  // namespace

void build_octree(const float* pos4, int n_seg, int seg_cap, const int* count, const float center[3], float width,
                  WorkerPool& pool, HostTree& out) {
    size_t n = 0;
    for (int s = 0; s < n_seg; ++s) n += size_t(count[s]);
    static thread_local std::vector<Item> buf_a, buf_b;
    static thread_local std::vector<uint8_t> code;
    buf_a.resize(n); buf_b.resize(n); code.resize(n);
    size_t k = 0;
    for (int s = 0; s < n_seg; ++s)
        for (int j = 0; j < count[s]; ++j, ++k) {
            const float* p = pos4 + 4 * (size_t(s) * seg_cap + j);
            buf_a[k] = Item{p[0], p[1], p[2], p[3], int32_t(s * seg_cap + j)};
        }
    Box root;
    root.c[0] = center[0]; root.c[1] = center[1]; root.c[2] = center[2];
    root.hw = width * 0.5f;  // Bounds::new, shared.rs:236-243
    root.w = width;

    std::vector<TopEntry> top;
    std::vector<Task> tasks;
    build_top(buf_a.data(), buf_b.data(), code.data(), int(n), root, 0, top, tasks);

    pool.run(int(tasks.size()), [&](int t) {
        Task& tk = tasks[t];
        tk.out.a.reserve(size_t(tk.n) * 2);
        tk.out.b.reserve(size_t(tk.n) * 2);
        tk.out.order.reserve(tk.n);
        build_rec(tk.src, tk.tmp, tk.code, tk.n, tk.box, tk.depth, tk.out);
    });

    // splice: final index of every top entry = prefix sum of entry sizes
    std::vector<int> first(top.size() + 1, 0);
    std::vector<int> ofirst(top.size() + 1, 0);
    for (size_t e = 0; e < top.size(); ++e) {
        int sz = top[e].task >= 0 ? int(tasks[top[e].task].out.a.size()) : 1;
        int osz = top[e].task >= 0 ? int(tasks[top[e].task].out.order.size()) : (top[e].b.body >= 0 ? 1 : 0);
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
            out.a[first[e]] = top[e].a;
            NodeB b = top[e].b;
            b.skip = first[top[e].end];
            out.b[first[e]] = b;
            if (b.body >= 0) out.order[ofirst[e]] = b.body;
        }
    }
    pool.run(int(top.size()), [&](int e) {
        if (top[e].task < 0) return;
        const Emit& em = tasks[top[e].task].out;
        const int base = first[e];
        std::memcpy(&out.a[base], em.a.data(), em.a.size() * sizeof(NodeA));
        for (size_t i = 0; i < em.b.size(); ++i) {
            NodeB b = em.b[i];
            b.skip += base;
            out.b[base + i] = b;
        }
        if (!em.order.empty()) std::memcpy(&out.order[ofirst[e]], em.order.data(), em.order.size() * sizeof(int32_t));
    });
    for (auto& tk : tasks) out.too_deep = out.too_deep || tk.out.too_deep;
}

}  // namespace nbody
