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

#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <array>

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
        stop_a_.store(true);
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
}

// Workers spin on the epoch counter for a while before they sleep on the condition variable: a
// Barnes-Hut step calls run() five times within ~2 ms, and a futex wake-up per call per worker
// (tens of microseconds each) would cost as much as the partition work itself.
void WorkerPool::loop() {
    uint64_t seen = 0;
    for (;;) {
        bool got = false;
        for (int spin = 0; spin < 4000; ++spin) {  // ~0.1 ms
            if (epoch_a_.load(std::memory_order_acquire) != seen || stop_a_.load(std::memory_order_relaxed)) { got = true; break; }
            __builtin_ia32_pause();
        }
        if (!got) {
            std::unique_lock<std::mutex> lk(m_);
            sleepers_++;
            cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
            sleepers_--;
        }
        if (stop_a_.load(std::memory_order_relaxed)) return;
        seen = epoch_a_.load(std::memory_order_acquire);
        const std::function<void(int)>* fn = fn_;
        const int n = n_tasks_;
        for (;;) {
            int t = next_.fetch_add(1, std::memory_order_relaxed);
            if (t >= n) break;
            (*fn)(t);
        }
        active_a_.fetch_sub(1, std::memory_order_acq_rel);
    }
}

void WorkerPool::run(int n_tasks, const std::function<void(int)>& fn) {
    if (n_tasks <= 0) return;
    if (workers_.empty() || n_tasks == 1) {
        for (int t = 0; t < n_tasks; ++t) fn(t);
        return;
    }
    bool wake;
    {
        std::lock_guard<std::mutex> lk(m_);
        fn_ = &fn;
        n_tasks_ = n_tasks;
        next_.store(0, std::memory_order_relaxed);
        active_a_.store(int(workers_.size()), std::memory_order_relaxed);
        ++epoch_;
        epoch_a_.store(epoch_, std::memory_order_release);
        wake = sleepers_ > 0;
    }
    if (wake) cv_.notify_all();
    for (;;) {  // the caller works too
        int t = next_.fetch_add(1, std::memory_order_relaxed);
        if (t >= n_tasks) break;
        fn(t);
    }
    while (active_a_.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();  // every worker has left fn
}

// ------------------------------------------------------------------------------ output arrays
void HostTree::clear() {
    auto rel = release ? release : +[](void* p) { std::free(p); };
    if (nodes) rel(nodes);
    if (order) rel(order);
    nodes = nullptr; order = nullptr;
    cap_nodes = cap_order = n_nodes = n_order = 0;
}

void HostTree::reserve(size_t n_nodes_wanted, size_t order_n) {
    auto al = alloc ? alloc : +[](size_t n) { return std::malloc(n); };
    auto rel = release ? release : +[](void* p) { std::free(p); };
    if (n_nodes_wanted > cap_nodes) {
        size_t cap = n_nodes_wanted + n_nodes_wanted / 4 + 64;
        if (nodes) rel(nodes);
        nodes = static_cast<NodeRec*>(al(cap * sizeof(NodeRec)));
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
    std::vector<NodeRec> nodes;
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
    const int me = int(e.nodes.size());
    e.nodes.push_back(NodeRec{NodeA{0.f, 0.f, 0.f, 0.f}, NodeB{box.w * box.w, me + 1, box.w, -1}});
    if (n == 0) return;
    if (n == 1) {
        e.nodes[me].a = NodeA{src[0].x, src[0].y, src[0].z, src[0].m};
        e.nodes[me].b.body = src[0].id;
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
    e.nodes[me].a = node;
    e.nodes[me].b.skip = int(e.nodes.size());
}

constexpr int kTaskDepth = 2;

struct Task { Item* src; Item* tmp; uint8_t* code; int n; Box box; int depth; Emit* out; };

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
        tasks.push_back(Task{src, tmp, code, n, box, depth, nullptr});
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

struct BuildScratch::Impl {
    std::vector<Item> buf_a, buf_b;
    std::vector<uint8_t> code;
    std::vector<Emit> emits;  // one per subtree task; capacities survive from step to step
};
BuildScratch::BuildScratch() : impl(new Impl) {}
BuildScratch::~BuildScratch() { delete impl; }

void build_octree(const float* pos4, int n_seg, int seg_cap, const int* count, const float center[3], float width,
                  WorkerPool& pool, BuildScratch& scratch, HostTree& out) {
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
    root.hw = width * 0.5f;  // Bounds::new, shared.rs:236-243
    root.w = width;

    const int T = pool.size();
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
            const float* p = pos4 + 4 * (size_t(s) * seg_cap + j);
            A[k] = Item{p[0], p[1], p[2], p[3], int32_t(size_t(s) * seg_cap + j)};
        }
    };

    std::vector<TopEntry> top;
    std::vector<Task> tasks;
    if (n < 8192 || T == 1) {
        fill(0, n);
        build_top(A, B, C, int(n), root, 0, top, tasks);
    } else {
        // ---- level 0 in parallel: chunked classify + stable scatter; the root's sums are one
        // sequential fold (their order is the reference's) running beside the chunk tasks
        const int NC = T;
        const size_t chunk = (n + NC - 1) / NC;
        std::vector<std::array<int, 8>> cnt(NC);
        NodeA root_node{0.f, 0.f, 0.f, 0.f};
        pool.run(NC, [&](int t) {
            const size_t k0 = std::min(n, size_t(t) * chunk), k1 = std::min(n, k0 + chunk);
            fill(k0, k1);
            std::array<int, 8> c{};
            for (size_t k = k0; k < k1; ++k) {
                const Item& it = A[k];
                int o = (it.x > root.c[0] ? 1 : 0) | (it.y > root.c[1] ? 2 : 0) | (it.z > root.c[2] ? 4 : 0);
                C[k] = uint8_t(o);
                c[o]++;
            }
            cnt[t] = c;
        });
        int start[8], tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = 0; t < NC; ++t) for (int o = 0; o < 8; ++o) tot[o] += cnt[t][o];
        { int run = 0; for (int o = 0; o < 8; ++o) { start[o] = run; run += tot[o]; } }
        std::vector<std::array<int, 8>> off(NC);
        { int run[8]; for (int o = 0; o < 8; ++o) run[o] = start[o];
          for (int t = 0; t < NC; ++t) for (int o = 0; o < 8; ++o) { off[t][o] = run[o]; run[o] += cnt[t][o]; } }
        pool.run(NC + 1, [&](int t) {
            if (t == NC) {  // mass = sum m, com = sum(pos*m)/mass, folded left to right over all bodies
                float mass = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
                for (size_t k = 0; k < n; ++k) {
                    const Item& it = A[k];
                    mass += it.m; sx += it.x * it.m; sy += it.y * it.m; sz += it.z * it.m;
                }
                root_node = NodeA{sx / mass, sy / mass, sz / mass, mass};
                return;
            }
            const size_t k0 = std::min(n, size_t(t) * chunk), k1 = std::min(n, k0 + chunk);
            int o8[8];
            for (int o = 0; o < 8; ++o) o8[o] = off[t][o];
            for (size_t k = k0; k < k1; ++k) B[o8[C[k]]++] = A[k];
        });
        // ---- level 1: one task per non-empty orthant; each registers its depth-2 subtrees
        std::vector<std::vector<TopEntry>> ctop(8);
        std::vector<std::vector<Task>> ctasks(8);
        pool.run(8, [&](int o) {
            if (tot[o]) build_top(B + start[o], A + start[o], C + start[o], tot[o], root.child(o), 1, ctop[o], ctasks[o]);
        });
        top.emplace_back();
        top[0].a = root_node;
        top[0].b = NodeB{root.w * root.w, 0, root.w, -1};
        for (int o = 0; o < 8; ++o) {
            const int ebase = int(top.size()), tbase = int(tasks.size());
            for (TopEntry e : ctop[o]) {
                if (e.task >= 0) e.task += tbase;
                e.end += ebase;
                top.push_back(e);
            }
            for (Task& t : ctasks[o]) tasks.push_back(t);
        }
        top[0].end = int(top.size());
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
        e.nodes.clear(); e.order.clear(); e.too_deep = false;
        build_rec(tk.src, tk.tmp, tk.code, tk.n, tk.box, tk.depth, e);
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
    for (auto& tk : tasks) out.too_deep = out.too_deep || tk.out->too_deep;
    lap("splice");
}

}  // namespace nbody
