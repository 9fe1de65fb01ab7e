// nbody_f64.cpp -- host orchestration of an F = f64 handle: the reference's `Simulation<f64, 3, PointParticle<f64,3>, _>`
// (the instantiation its own driver uses, src/main.rs:52-105).  Strict arithmetic with the octree built on the host in f64
// (octree_host.cpp, the same stable 8-way partition as for f32): positions, velocities, accelerations and node counts
// equal the oracle's f64 instantiation bit for bit -- on one shard and over index-block shards alike (the blocks'
// positions and live counts are exchanged once per step through the handle's transport; partners and tree bodies keep
// their global order).  NBODY_MATH_FAST: the fast walk (one running sum per lane, split node range), device build on one
// shard.  Bodies cross the boundary as 80-byte records.
#include "nbody_f64.h"
#include "kernels_f64.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

namespace nbody64 {

struct State {
    Dev d;
    double g = 1.0, g_soft = 0.0, dt = 1e-3, theta2 = 0.5;   // shared.rs:69-78
    double center[3] = {0.0, 0.0, 0.0};
    double width = 0.0;
    Bounds64 bnd{};
    bool bounds_set = false;
    double elapsed = 0.0;
    size_t n_local = 0;        // host view of the body count (an upper bound while count_dirty)
    bool count_dirty = false;
    int* h_count = nullptr;    // pinned [n_seg + 1]: the blocks' live counts
    std::vector<int> count_upper;   // host bound of every block's live count (counts only shrink between uploads)
    std::vector<int> own_order;
    double* d_aos = nullptr;   // staging for PointParticle<f64,3> records
    double* h_aos = nullptr;   // pinned
    size_t aos_cap = 0;
    // Barnes-Hut
    nbody::HostTreeT<double> tree;
    nbody::BuildScratchT<double> scratch;
    Node64* d_nodes = nullptr;
    size_t node_cap = 0;
    int* d_order = nullptr;
    size_t order_cap = 0;
    double* h_pos = nullptr;   // pinned [4 * cap]
    Open64* d_stack = nullptr; // nested walk: [levels][lanes]
    size_t stack_lanes = 0;
    int stack_levels = 0;
    double* d_energy = nullptr;
    size_t energy_blocks = 0;
    // device-side build (NBODY_TREE_DEVICE): kernels_tree.hip instantiated for double
    void* d_tree_ws = nullptr;
    const double* kick_dt = nullptr;   // inside a step: the dt the force pass may apply itself (fast walk, split node range)
    int kicked = 0;
    void* d_tree_cat = nullptr;   // sharded worlds, device build: the gathered live bodies + the own-order filter's arrays
    size_t tree_ws_cap = 0;    // bodies the workspace is sized for
    int* d_tree_info = nullptr;   // [4] {nodes, flags, bodies}
    int* h_tree_info = nullptr;   // pinned
    nbody::TreeDevWork tree_work;
    bool tree_on_device = false;  // where the last tree lives (nbody_tree_export)
    size_t dev_nodes = 0;
    // fast walk (NBODY_MATH_FAST): node-range split points and the segments' partial sums
    int* d_split = nullptr;       // [kMaxSplit + 1 + kMaxSplit + kMaxSplit * kMaxAnc64] first[], n_anc[], anc[][]
    int* h_split = nullptr;       // pinned mirror (host-built tree)
    double4* d_planes = nullptr;  // [K][cap]
    size_t planes_cap = 0;
};

namespace {

using clk = std::chrono::steady_clock;
inline double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

int fail(NbodyHandle* h, int code, const std::string& msg) {
    h->err = msg;
    return code;
}

#define HIP_TRY(h, expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(h, NBODY_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

void* pinned_alloc(size_t n) {
    void* p = nullptr;
    if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void pinned_free(void* p) { (void)hipHostFree(p); }

int ensure_aos(NbodyHandle* h, State& s, size_t records) {
    if (records <= s.aos_cap) return NBODY_OK;
    if (s.d_aos) (void)hipFree(s.d_aos);
    if (s.h_aos) (void)hipHostFree(s.h_aos);
    s.d_aos = nullptr; s.h_aos = nullptr; s.aos_cap = 0;
    HIP_TRY(h, hipMalloc(&s.d_aos, records * 10 * sizeof(double)));
    HIP_TRY(h, hipHostMalloc(&s.h_aos, records * 10 * sizeof(double), hipHostMallocDefault));
    s.aos_cap = records;
    return NBODY_OK;
}

int sync_count(NbodyHandle* h, State& s) {
    if (!s.count_dirty) return NBODY_OK;
    HIP_TRY(h, hipMemcpyAsync(s.h_count, s.d.seg_count, sizeof(int) * s.d.n_seg, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int g = 0; g < s.d.n_seg; ++g) s.count_upper[size_t(g)] = s.h_count[g];
    s.n_local = size_t(s.h_count[s.d.my_seg]);
    s.count_dirty = false;
    return NBODY_OK;
}

int push_count(NbodyHandle* h, State& s) {
    s.h_count[s.d.n_seg] = int(s.n_local);
    s.count_upper[size_t(s.d.my_seg)] = int(s.n_local);
    HIP_TRY(h, hipMemcpyAsync(s.d.count, s.h_count + s.d.n_seg, sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return NBODY_OK;
}

struct ForceTimer {  // HIP events around a force-kernel launch, on the launch stream (as in nbody_api.cpp)
    NbodyHandle* h;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    explicit ForceTimer(NbodyHandle* hh) : h(hh) {
        if (!h->profiling) return;
        if (!h->ev_free.empty()) { ev = h->ev_free.back(); h->ev_free.pop_back(); }
        else if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) { ev = {nullptr, nullptr}; return; }
        (void)hipEventRecord(ev.first, h->stream);
    }
    ~ForceTimer() {
        if (!ev.first) return;
        (void)hipEventRecord(ev.second, h->stream);
        h->ev_pending.push_back(ev);
    }
};

int bf_forces(NbodyHandle* h, State& s) {
    const double eps2 = s.g_soft * s.g_soft;  // brute_force.rs:69
    {
        ForceTimer t(h);
        launch_bf_strict(h->stream, s.d, int(s.n_local), s.g, eps2);
    }
    HIP_TRY(h, hipGetLastError());
    if (h->profiling && s.n_local > 0) {
        uint64_t tot = 0;
        for (int c : s.count_upper) tot += uint64_t(c);
        h->stats.force_kernel_interactions += uint64_t(s.n_local) * (tot - 1);
    }
    return NBODY_OK;
}

int ensure_stack(NbodyHandle* h, State& s, int levels) {   // the nested sums' stack: one entry per open cell on the lane's path
    const size_t lanes = (size_t(s.d.cap) + 255) / 256 * 256;
    if (lanes > s.stack_lanes || levels > s.stack_levels) {
        if (s.d_stack) (void)hipFree(s.d_stack);
        s.d_stack = nullptr; s.stack_lanes = 0; s.stack_levels = 0;
        const int lv = std::max(levels + 8, 32);
        HIP_TRY(h, hipMalloc(&s.d_stack, lanes * size_t(lv) * sizeof(Open64)));
        s.stack_lanes = lanes; s.stack_levels = lv;
    }
    return NBODY_OK;
}

// The fast walk (NBODY_MATH_FAST on an f64 handle): one running sum per lane, node range split over K segments so that a
// few ten thousand bodies still fill the chip.  `host_nodes` != nullptr: the split points' ancestors are listed here from
// the host-built tree; nullptr: by k_tree_split_anc from the device build's arrays (n_tree bodies).
constexpr int kMaxSplit = 64;
int fast_walk(NbodyHandle* h, State& s, const Node64* nodes, int n_nodes, const int* order, int n_order, const nbody::NodeRecT<double>* host_nodes, int n_tree,
              int k_done = 0 /* > 0: the split points of this many segments are on the device already (they rode in the build) */) {
    constexpr size_t kSplitInts = kMaxSplit + 1 + kMaxSplit + size_t(kMaxSplit) * kMaxAnc64;
    if (!s.d_split) {
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&s.d_split), kSplitInts * sizeof(int)));
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&s.h_split), kSplitInts * sizeof(int), hipHostMallocDefault));
    }
    const nbody::WalkPlan plan = nbody::walk_plan(size_t(n_order), true, kMaxSplit, float(s.theta2));   // (bodies per lane x segments: kernels.h)
    int K = plan.segments;
    while (nbody::tuning().bh_walk_split <= 0 && K > 1 && K * 16 > n_nodes) K /= 2;
    if (k_done > 0) K = k_done;
    if (n_order == 0 || n_nodes <= 0) return NBODY_OK;
    int* first = s.d_split;
    int* n_anc = s.d_split + kMaxSplit + 1;
    int* anc = s.d_split + kMaxSplit + 1 + kMaxSplit;
    if (k_done > 0) {
    } else if (host_nodes) {
        int* hf = s.h_split;
        int* hn = hf + kMaxSplit + 1;
        int* ha = hn + kMaxSplit;
        for (int k = 0; k <= K; ++k) hf[k] = int((long long)n_nodes * k / K);
        for (int k = 0; k < K; ++k) {   // ancestors of first[k]: down from the root along the skip links
            int cnt = 0, j = 0;
            const int target = hf[k];
            while (j != target && cnt < kMaxAnc64) {
                ha[k * kMaxAnc64 + cnt++] = j;
                int c = j + 1;
                while (host_nodes[c].b.skip <= target) c = host_nodes[c].b.skip;
                j = c;
            }
            hn[k] = cnt;
        }
        HIP_TRY(h, hipMemcpyAsync(s.d_split, s.h_split, (kMaxSplit + 1 + kMaxSplit + size_t(K) * kMaxAnc64) * sizeof(int), hipMemcpyHostToDevice, h->stream));
    } else {
        nbody::launch_tree_split_anc(h->stream, s.tree_work, n_tree, n_nodes, K, first, n_anc, anc, kMaxAnc64);
    }
    if (K > 1 && size_t(K) * size_t(s.d.cap) > s.planes_cap) {
        if (s.d_planes) (void)hipFree(s.d_planes);
        s.d_planes = nullptr; s.planes_cap = 0;
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&s.d_planes), size_t(K) * size_t(s.d.cap) * sizeof(double4)));
        s.planes_cap = size_t(K) * size_t(s.d.cap);
    }
    WalkSplit64 sp{K, first, anc, n_anc, s.d_planes, size_t(s.d.cap)};
    {
        ForceTimer t(h);
        launch_bh_walk_fast(h->stream, s.d, nodes, n_nodes, order, n_order, s.g, s.g_soft * s.g_soft, s.theta2, h->d_counters,
                            h->cfg.leaf_mode == NBODY_LEAF_DIRECT ? 1 : 0, sp, std::min(3, plan.bodies_per_lane), s.kick_dt, &s.kicked);   // (64-byte records, doubles in registers: beyond three per lane the f64 walk loses again -- tools/f64_walk_probe.py)
    }
    HIP_TRY(h, hipGetLastError());
    return NBODY_OK;
}

// The tree built on the device (NBODY_TREE_DEVICE; kernels_tree.hip for double): no positions to the host, no nodes
// back.  Same cells, pre-order and skip links as the host build; centres of mass from f64 prefix sums instead of the
// reference's sequential f64 folds (last bits).  *fell_back: coincident bodies / > 42 levels -> the caller builds on the host.
int bh_forces_device(NbodyHandle* h, State& s, bool* fell_back) {
    *fell_back = false;
    auto t0 = clk::now();
    const int G = s.d.n_seg;
    const bool sharded = G > 1;   // every rank builds the world's tree from the gathered positions (fast math; kernels_tree.hip k_tree_cat64)
    const size_t cap = size_t(s.d.cap) * size_t(G);
    size_t tot_upper = 0;
    for (int g = 0; g < G; ++g) tot_upper += size_t(s.count_upper[size_t(g)]);
    if (!sharded) tot_upper = s.n_local;
    if (s.tree_ws_cap < cap) {
        if (s.d_tree_ws) (void)hipFree(s.d_tree_ws);
        if (s.d_tree_cat) (void)hipFree(s.d_tree_cat);
        s.d_tree_ws = nullptr; s.d_tree_cat = nullptr; s.tree_ws_cap = 0;
        HIP_TRY(h, hipMalloc(&s.d_tree_ws, nbody::tree_build_workspace_bytes(cap)));
        if (sharded) HIP_TRY(h, hipMalloc(&s.d_tree_cat, nbody::tree_cat_bytes64(cap)));
        s.tree_ws_cap = cap;
    }
    if (!s.d_tree_info) {
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&s.d_tree_info), 4 * sizeof(int)));
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&s.h_tree_info), 4 * sizeof(int), hipHostMallocDefault));
    }
    if (cap > s.order_cap) {
        if (s.d_order) (void)hipFree(s.d_order);
        s.d_order = nullptr; s.order_cap = 0;
        HIP_TRY(h, hipMalloc(&s.d_order, cap * sizeof(int)));
        s.order_cap = cap;
    }
    const double4* tree_pos = s.d.pos;
    const int* tree_count = s.d.count;
    nbody::TreeCat cat;
    if (sharded) {
        double4* pos_cat = nullptr;
        cat = nbody::tree_cat_layout64(s.d_tree_cat, cap, &pos_cat);
        nbody::launch_tree_cat64(h->stream, s.d.pos_all, s.d.seg_count, G, s.d.cap, s.d.my_seg, pos_cat, cat.info);
        tree_pos = pos_cat;
        tree_count = cat.info;
    }
    // one shard, fast math: the walk's split points ride in the build's last launch (kernels.h TreeSplitReq)
    nbody::TreeSplitReq req;
    int k_pre = 0;
    if (!sharded && h->cfg.math_mode == NBODY_MATH_FAST && tot_upper > 0) {
        constexpr size_t kSplitInts = kMaxSplit + 1 + kMaxSplit + size_t(kMaxSplit) * kMaxAnc64;
        if (!s.d_split) {
            HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&s.d_split), kSplitInts * sizeof(int)));
            HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&s.h_split), kSplitInts * sizeof(int), hipHostMallocDefault));
        }
        int K = nbody::walk_plan(tot_upper, true, kMaxSplit, float(s.theta2)).segments;
        while (nbody::tuning().bh_walk_split <= 0 && K > 1 && size_t(K) * 16 > tot_upper) K /= 2;   // (a tree has at least as many nodes as bodies)
        req.n_split = K; req.first = s.d_split; req.n_anc = s.d_split + kMaxSplit + 1; req.anc = s.d_split + kMaxSplit + 1 + kMaxSplit;
        req.max_anc = kMaxAnc64; req.info = s.d_tree_info; req.poison = nullptr;
        k_pre = K;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (s.node_cap < 2 * tot_upper + 64) {
            if (s.d_nodes) (void)hipFree(s.d_nodes);
            s.d_nodes = nullptr; s.node_cap = 0;
            const size_t want = std::max<size_t>(2 * tot_upper + 64, s.dev_nodes + s.dev_nodes / 4 + 1024);
            HIP_TRY(h, hipMalloc(&s.d_nodes, want * sizeof(Node64)));
            s.node_cap = want;
        }
        if (nbody::build_octree_device_f64(h->stream, tree_pos, tree_count, int(tot_upper), s.center, s.width, s.d_tree_ws, s.tree_ws_cap, s.d_nodes,
                                           int(std::min<size_t>(s.node_cap, 0x7fffffff)), s.d_order, s.d_tree_info, &s.tree_work, k_pre ? &req : nullptr) != 0)
            return fail(h, NBODY_ERR_HIP, "device octree build: rocPRIM call failed");
        HIP_TRY(h, hipMemcpyAsync(s.h_tree_info, s.d_tree_info, 3 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        if (sharded) HIP_TRY(h, hipMemcpyAsync(s.h_count, s.d.seg_count, sizeof(int) * size_t(G), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (s.h_tree_info[1] & 5) { *fell_back = true; return NBODY_OK; }
        if (!(s.h_tree_info[1] & 2)) break;
        s.dev_nodes = size_t(s.h_tree_info[0]);   // the array was too small: the build says how many it needs
        s.node_cap = 0;
        if (attempt == 1) return fail(h, NBODY_ERR_CAPACITY, "device octree build: node array too small twice");
    }
    if (sharded) {   // the live counts of every block, and the own bodies' places in the tree order
        size_t total = 0;
        for (int g = 0; g < G; ++g) { s.count_upper[size_t(g)] = s.h_count[g]; total += size_t(s.h_count[g]); }
        s.dev_nodes = size_t(s.h_tree_info[0]);
        s.n_local = size_t(s.h_count[s.d.my_seg]);
        s.count_dirty = false;
        s.tree_on_device = true;
        if (nbody::launch_tree_own_order(h->stream, s.d_order, cat, int(total), s.d_tree_ws, nbody::tree_build_tmp_bytes(cap)) != 0)
            return fail(h, NBODY_ERR_HIP, "device octree build: rocPRIM call failed");
        h->stats.tree_build_ms += ms_since(t0);
        h->stats.tree_nodes = s.dev_nodes;
        return fast_walk(h, s, s.d_nodes, int(s.dev_nodes), cat.own_order, int(s.n_local), nullptr, int(total));
    }
    s.dev_nodes = size_t(s.h_tree_info[0]);
    s.n_local = size_t(s.h_tree_info[2]);
    s.count_dirty = false;
    s.tree_on_device = true;
    h->stats.tree_build_ms += ms_since(t0);
    h->stats.tree_nodes = s.dev_nodes;
    if (h->cfg.math_mode == NBODY_MATH_FAST) return fast_walk(h, s, s.d_nodes, int(s.dev_nodes), s.d_order, int(s.n_local), nullptr, int(s.n_local), k_pre);
    const bool direct = h->cfg.leaf_mode == NBODY_LEAF_DIRECT;
    if (!direct) { int rc = ensure_stack(h, s, 45); if (rc) return rc; }   // (the device build goes to 42 levels)
    {
        ForceTimer t(h);
        launch_bh_walk(h->stream, s.d, s.d_nodes, int(s.dev_nodes), s.d_order, int(s.n_local), s.g, s.g_soft * s.g_soft, s.theta2, h->d_counters,
                       direct ? 1 : 0, s.d_stack, s.stack_lanes);
    }
    HIP_TRY(h, hipGetLastError());
    return NBODY_OK;
}

// BarnesHutSimulation::update_forces (barnes_hut.rs:250-263): rebuild the tree (host, f64), one walk per body
int bh_forces(NbodyHandle* h, State& s) {
    if (h->cfg.tree_build == NBODY_TREE_DEVICE && (s.d.n_seg == 1 || h->cfg.math_mode == NBODY_MATH_FAST)) {   // (a sharded world: fast math only, create says so)
        bool fell_back = false;
        int rc = bh_forces_device(h, s, &fell_back);
        if (rc || !fell_back) return rc;
    }
    s.tree_on_device = false;
    auto t0 = clk::now();
    const int G = s.d.n_seg;
    for (int g = 0; g < G; ++g) {   // every block's positions (upper-bound counts) + the live counts, one synchronisation
        const size_t cnt = size_t(s.count_upper[size_t(g)]);
        if (cnt) HIP_TRY(h, hipMemcpyAsync(s.h_pos + 4 * size_t(g) * s.d.cap, s.d.pos_all + size_t(g) * s.d.cap, cnt * sizeof(double4), hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, hipMemcpyAsync(s.h_count, s.d.seg_count, sizeof(int) * G, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int g = 0; g < G; ++g) s.count_upper[size_t(g)] = s.h_count[g];
    s.n_local = size_t(s.h_count[s.d.my_seg]);
    s.count_dirty = false;
    const double copy_ms = ms_since(t0);
    auto t1 = clk::now();
    nbody::build_octree<double>(s.h_pos, G, s.d.cap, s.h_count, s.center, s.width, *h->pool, s.scratch, s.tree);
    if (s.tree.too_deep) return fail(h, NBODY_ERR_TREE_DEPTH, "octree deeper than NBODY_MAX_TREE_DEPTH (coincident bodies?)");
    // the own bodies in tree order (ids are block * cap + index in the block)
    const int32_t* order = s.tree.order;
    size_t n_order = s.tree.n_order;
    if (G > 1) {
        s.own_order.clear();
        const int lo = s.d.my_seg * s.d.cap, hi = lo + s.d.cap;
        for (size_t k = 0; k < s.tree.n_order; ++k) {
            const int id = s.tree.order[k];
            if (id >= lo && id < hi) s.own_order.push_back(id - lo);
        }
        order = s.own_order.data();
        n_order = s.own_order.size();
    }
    h->stats.tree_build_ms += ms_since(t1);
    h->stats.tree_nodes = s.tree.n_nodes;
    auto t2 = clk::now();
    if (s.tree.n_nodes > s.node_cap) {
        if (s.d_nodes) (void)hipFree(s.d_nodes);
        s.d_nodes = nullptr; s.node_cap = 0;
        const size_t cap = s.tree.n_nodes + s.tree.n_nodes / 4 + 1024;
        HIP_TRY(h, hipMalloc(&s.d_nodes, cap * sizeof(Node64)));
        s.node_cap = cap;
    }
    if (n_order > s.order_cap) {
        if (s.d_order) (void)hipFree(s.d_order);
        s.d_order = nullptr; s.order_cap = 0;
        const size_t cap = n_order + n_order / 4 + 1024;
        HIP_TRY(h, hipMalloc(&s.d_order, cap * sizeof(int)));
        s.order_cap = cap;
    }
    static_assert(sizeof(nbody::NodeRecT<double>) == sizeof(Node64), "host and device node records must agree");
    HIP_TRY(h, hipMemcpyAsync(s.d_nodes, s.tree.nodes, s.tree.n_nodes * sizeof(Node64), hipMemcpyHostToDevice, h->stream));
    if (n_order) HIP_TRY(h, hipMemcpyAsync(s.d_order, order, n_order * sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (G > 1) HIP_TRY(h, hipStreamSynchronize(h->stream));   // own_order is pageable and reused
    h->stats.tree_copy_ms += copy_ms + ms_since(t2);
    if (h->cfg.math_mode == NBODY_MATH_FAST)
        return fast_walk(h, s, s.d_nodes, int(s.tree.n_nodes), s.d_order, int(n_order), s.tree.nodes, int(s.tree.n_order));
    const bool direct = h->cfg.leaf_mode == NBODY_LEAF_DIRECT;
    if (!direct) { int rc = ensure_stack(h, s, s.tree.max_depth + 2); if (rc) return rc; }   // the tree's depth
    {
        ForceTimer t(h);
        launch_bh_walk(h->stream, s.d, s.d_nodes, int(s.tree.n_nodes), s.d_order, int(n_order), s.g, s.g_soft * s.g_soft, s.theta2,
                       h->d_counters, direct ? 1 : 0, s.d_stack, s.stack_lanes);
    }
    HIP_TRY(h, hipGetLastError());
    return NBODY_OK;
}

int forces(NbodyHandle* h, State& s) { return h->cfg.method == NBODY_BARNES_HUT ? bh_forces(h, s) : bf_forces(h, s); }

// index-block shards: the once-per-step exchange, in place, on the handle's stream
int exchange(NbodyHandle* h, State& s) {
    if (s.d.n_seg == 1 && !h->comm_ready) return NBODY_OK;
    if (!h->comm_ready) return fail(h, NBODY_ERR_COMM, "world_size > 1 but nbody_comm_init has not been called");
    int rc = h->tp->group_begin();
    if (!rc) rc = h->tp->all_gather(s.d.pos_all, size_t(s.d.cap) * sizeof(double4), h->stream);
    if (!rc) rc = h->tp->all_gather(s.d.seg_count, sizeof(int), h->stream);
    if (!rc) rc = h->tp->group_end();
    if (rc) return fail(h, rc, "f64 exchange: " + h->tp->error());
    return NBODY_OK;
}

int step_impl(NbodyHandle* h, State& s, double dt) {
    if (!s.bounds_set) return fail(h, NBODY_ERR_INVALID, "nbody_set_bounds has not been called");
    launch_drift_half(h->stream, s.d, int(s.n_local), dt, s.bnd);   // integrate_pre_force
    launch_compact(h->stream, s.d, int(s.n_local));                 // retain
    s.count_dirty = true;
    HIP_TRY(h, hipGetLastError());
    int rc = exchange(h, s);                                        // sharded: every block's positions and live count
    if (rc) return rc;
    s.kick_dt = &dt; s.kicked = 0;                                  // (the fast walk's plane reduction can take the kick along)
    rc = forces(h, s);                                              // update_forces
    s.kick_dt = nullptr;
    if (rc) return rc;
    if (!s.kicked) launch_kick_drift(h->stream, s.d, int(s.n_local), dt);   // integrate_after_force
    HIP_TRY(h, hipGetLastError());
    s.elapsed += dt;                                                // elapsed += dt
    h->stats.steps += 1;
    return NBODY_OK;
}

}  // namespace

int create(NbodyHandle* h) {
    State* sp = new State();
    h->f64 = sp;
    State& s = *sp;
    const int G = h->cfg.world_size;
    const size_t cap = (size_t(h->cfg.capacity) + size_t(G) - 1) / size_t(G);   // bodies a block can hold
    s.d.cap = int(cap);
    s.d.n_seg = G;
    s.d.my_seg = h->cfg.rank;
    s.count_upper.assign(size_t(G), 0);
    HIP_TRY(h, hipMalloc(&s.d.pos_all, size_t(G) * cap * sizeof(double4)));
    s.d.pos = s.d.pos_all + size_t(s.d.my_seg) * cap;
    HIP_TRY(h, hipMalloc(&s.d.vel, cap * sizeof(double4)));
    HIP_TRY(h, hipMalloc(&s.d.acc, cap * sizeof(double4)));
    HIP_TRY(h, hipMalloc(&s.d.seg_count, sizeof(int) * G));
    s.d.count = s.d.seg_count + s.d.my_seg;
    HIP_TRY(h, hipMalloc(&s.d.escaped, sizeof(int)));
    HIP_TRY(h, hipMalloc(&s.d.keep, cap));
    const size_t tiles = (cap + 1023) / 1024 + 1;
    HIP_TRY(h, hipMalloc(&s.d.tile_state, tiles * sizeof(unsigned long long)));
    HIP_TRY(h, hipMalloc(&s.d.epoch, sizeof(int)));
    HIP_TRY(h, hipMalloc(&s.d.inter, sizeof(unsigned long long)));
    HIP_TRY(h, hipMemsetAsync(s.d.pos_all, 0, size_t(G) * cap * sizeof(double4), h->stream));
    HIP_TRY(h, hipMemsetAsync(s.d.vel, 0, cap * sizeof(double4), h->stream));
    HIP_TRY(h, hipMemsetAsync(s.d.acc, 0, cap * sizeof(double4), h->stream));
    HIP_TRY(h, hipMemsetAsync(s.d.seg_count, 0, sizeof(int) * G, h->stream));
    HIP_TRY(h, hipMemsetAsync(s.d.escaped, 0, sizeof(int), h->stream));
    HIP_TRY(h, hipMemsetAsync(s.d.keep, 1, cap, h->stream));
    HIP_TRY(h, hipMemsetAsync(s.d.tile_state, 0, tiles * sizeof(unsigned long long), h->stream));
    HIP_TRY(h, hipMemsetAsync(s.d.epoch, 0, sizeof(int), h->stream));
    HIP_TRY(h, hipMemsetAsync(s.d.epoch, 1, 1, h->stream));   // epoch = 1
    HIP_TRY(h, hipMemsetAsync(s.d.inter, 0, sizeof(unsigned long long), h->stream));
    HIP_TRY(h, hipHostMalloc(&s.h_count, (size_t(G) + 1) * sizeof(int), hipHostMallocDefault));
    if (h->cfg.method == NBODY_BARNES_HUT) {
        s.tree.alloc = pinned_alloc;
        s.tree.release = pinned_free;
        HIP_TRY(h, hipHostMalloc(&s.h_pos, size_t(G) * cap * sizeof(double4), hipHostMallocDefault));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return NBODY_OK;
}

void destroy(NbodyHandle* h) {
    State* s = h->f64;
    if (!s) return;
    s->tree.clear();
    void* dev[] = {s->d.pos_all, s->d.vel, s->d.acc, s->d.seg_count, s->d.escaped, s->d.keep, s->d.tile_state, s->d.epoch, s->d.inter,
                   s->d_aos, s->d_nodes, s->d_order, s->d_stack, s->d_energy, s->d_tree_ws, s->d_tree_cat, s->d_tree_info, s->d_split, s->d_planes};
    for (void* p : dev) if (p) (void)hipFree(p);
    void* host[] = {s->h_count, s->h_aos, s->h_pos, s->h_tree_info, s->h_split};
    for (void* p : host) if (p) (void)hipHostFree(p);
    delete s;
    h->f64 = nullptr;
}

int clone_state(NbodyHandle* src, NbodyHandle* dst) {
    State& a = *src->f64;
    State& b = *dst->f64;
    int rc = sync_count(src, a);
    if (rc) return rc;
    const size_t cap = size_t(a.d.cap);
    HIP_TRY(dst, hipStreamSynchronize(src->stream));
    HIP_TRY(dst, hipMemcpyAsync(b.d.pos_all, a.d.pos_all, size_t(a.d.n_seg) * cap * sizeof(double4), hipMemcpyDeviceToDevice, dst->stream));
    HIP_TRY(dst, hipMemcpyAsync(b.d.vel, a.d.vel, cap * sizeof(double4), hipMemcpyDeviceToDevice, dst->stream));
    HIP_TRY(dst, hipMemcpyAsync(b.d.acc, a.d.acc, cap * sizeof(double4), hipMemcpyDeviceToDevice, dst->stream));
    HIP_TRY(dst, hipMemcpyAsync(b.d.seg_count, a.d.seg_count, sizeof(int) * a.d.n_seg, hipMemcpyDeviceToDevice, dst->stream));
    b.count_upper = a.count_upper;
    HIP_TRY(dst, hipStreamSynchronize(dst->stream));
    b.g = a.g; b.g_soft = a.g_soft; b.dt = a.dt; b.theta2 = a.theta2;
    std::memcpy(b.center, a.center, sizeof(b.center));
    b.width = a.width; b.bnd = a.bnd; b.bounds_set = a.bounds_set;
    b.elapsed = a.elapsed;
    b.n_local = a.n_local;
    return NBODY_OK;   // (like the reference's BH clone, barnes_hut.rs:113-135, the tree is not carried over)
}

int upload(NbodyHandle* h, const void* aos, size_t n, size_t stride) {
    State& s = *h->f64;
    if (stride < 80 || stride % 8) return fail(h, NBODY_ERR_INVALID, "f64 handle: stride must be a multiple of 8 and >= 80 bytes");
    if (n > size_t(h->cfg.capacity)) return fail(h, NBODY_ERR_CAPACITY, "more bodies than NbodyConfig.capacity");
    int rc = ensure_aos(h, s, n);
    if (rc) return rc;
    const char* src = static_cast<const char*>(aos);
    for (size_t k = 0; k < n; ++k) std::memcpy(s.h_aos + 10 * k, src + k * stride, 80);
    if (n) HIP_TRY(h, hipMemcpyAsync(s.d_aos, s.h_aos, n * 80, hipMemcpyHostToDevice, h->stream));
    const size_t G = size_t(s.d.n_seg), blk = (n + G - 1) / G;   // contiguous index blocks keep the ascending-partner order
    for (size_t g = 0; g < G; ++g) {
        const size_t lo = std::min(n, g * blk), hi = std::min(n, lo + blk);
        s.count_upper[g] = int(hi - lo);
        s.h_count[g] = int(hi - lo);
        if (int(g) == s.d.my_seg) {
            launch_aos_to_soa(h->stream, s.d_aos + 10 * lo, 10, int(hi - lo), s.d, 0);
            s.n_local = hi - lo;
            h->first_global = lo; h->n_at_upload = hi - lo;
        } else {
            launch_aos_to_pos(h->stream, s.d_aos + 10 * lo, 10, int(hi - lo), s.d.pos_all + g * size_t(s.d.cap));
        }
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemsetAsync(s.d.escaped, 0, sizeof(int), h->stream));
    HIP_TRY(h, hipMemcpyAsync(s.d.seg_count, s.h_count, sizeof(int) * G, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    s.count_dirty = false;
    return NBODY_OK;
}

int download(NbodyHandle* h, void* aos, size_t cap, size_t stride, size_t* n_out) {
    State& s = *h->f64;
    if (stride < 80 || stride % 8) return fail(h, NBODY_ERR_INVALID, "f64 handle: stride must be a multiple of 8 and >= 80 bytes");
    int rc = sync_count(h, s);
    if (rc) return rc;
    const size_t n = s.n_local;
    if (n_out) *n_out = n;
    if (n > cap) return fail(h, NBODY_ERR_CAPACITY, "download buffer too small");
    if (n == 0) return NBODY_OK;
    if (!aos) return fail(h, NBODY_ERR_INVALID, "null buffer");
    rc = ensure_aos(h, s, n);
    if (rc) return rc;
    launch_soa_to_aos(h->stream, s.d_aos, 10, int(n), s.d);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(s.h_aos, s.d_aos, n * 80, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    char* dst = static_cast<char*>(aos);
    for (size_t k = 0; k < n; ++k) std::memcpy(dst + k * stride, s.h_aos + 10 * k, 80);
    return NBODY_OK;
}

int count(NbodyHandle* h, size_t* n_out) {
    State& s = *h->f64;
    int rc = sync_count(h, s);
    if (rc) return rc;
    *n_out = s.n_local;
    return NBODY_OK;
}

int count_global(NbodyHandle* h, size_t* n_out) {   // as of the last exchange
    State& s = *h->f64;
    s.count_dirty = true;
    int rc = sync_count(h, s);
    if (rc) return rc;
    size_t t = 0;
    for (int c : s.count_upper) t += size_t(c);
    *n_out = t;
    return NBODY_OK;
}

int add_point(NbodyHandle* h, const void* particle) {   // Vec::push (brute_force.rs:92-94)
    State& s = *h->f64;
    if (s.d.n_seg > 1) return fail(h, NBODY_ERR_INVALID, "add_point on a sharded f64 world is not supported (f32 handles: collective push / swap_remove)");
    int rc = sync_count(h, s);
    if (rc) return rc;
    if (s.n_local >= size_t(s.d.cap)) return fail(h, NBODY_ERR_CAPACITY, "capacity exhausted");
    rc = ensure_aos(h, s, 1);
    if (rc) return rc;
    std::memcpy(s.h_aos, particle, 80);
    HIP_TRY(h, hipMemcpyAsync(s.d_aos, s.h_aos, 80, hipMemcpyHostToDevice, h->stream));
    launch_aos_to_soa(h->stream, s.d_aos, 10, 1, s.d, s.n_local);
    HIP_TRY(h, hipGetLastError());
    s.n_local += 1;
    return push_count(h, s);
}

int remove_point(NbodyHandle* h, size_t index) {   // Vec::swap_remove (brute_force.rs:96-98)
    State& s = *h->f64;
    if (s.d.n_seg > 1) return fail(h, NBODY_ERR_INVALID, "remove_point on a sharded f64 world is not supported (f32 handles: collective push / swap_remove)");
    int rc = sync_count(h, s);
    if (rc) return rc;
    if (index >= s.n_local) return fail(h, NBODY_ERR_INVALID, "swap_remove index out of range");
    const size_t last = s.n_local - 1;
    if (index != last) {
        HIP_TRY(h, hipMemcpyAsync(s.d.pos + index, s.d.pos + last, sizeof(double4), hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(s.d.vel + index, s.d.vel + last, sizeof(double4), hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(s.d.acc + index, s.d.acc + last, sizeof(double4), hipMemcpyDeviceToDevice, h->stream));
    }
    s.n_local = last;
    return push_count(h, s);
}

int set_settings(NbodyHandle* h, double g, double g_soft, double dt, double theta2) {
    State& s = *h->f64;
    s.g = g; s.g_soft = g_soft; s.dt = dt; s.theta2 = theta2;
    return NBODY_OK;
}

int get_settings(const NbodyHandle* h, double* g, double* g_soft, double* dt, double* theta2) {
    const State& s = *h->f64;
    if (g) *g = s.g;
    if (g_soft) *g_soft = s.g_soft;
    if (dt) *dt = s.dt;
    if (theta2) *theta2 = s.theta2;
    return NBODY_OK;
}

int set_bounds(NbodyHandle* h, const double center[3], double width) {
    State& s = *h->f64;
    std::memcpy(s.center, center, sizeof(s.center));
    s.width = width;
    const double hw = width * 0.5;  // Bounds::new
    for (int i = 0; i < 3; ++i) {
        s.bnd.lo[i] = center[i] + (-hw);  // add_scalar(-half_width), shared.rs:224
        s.bnd.hi[i] = center[i] + hw;     // shared.rs:228
    }
    s.bounds_set = true;
    return NBODY_OK;
}

void get_bounds(const NbodyHandle* h, double center[3], double* width) {
    const State& s = *h->f64;
    std::memcpy(center, s.center, sizeof(s.center));
    *width = s.width;
}

int init(NbodyHandle* h) {
    h->f64->elapsed = 0.0;
    return NBODY_OK;
}

int step_by(NbodyHandle* h, double dt) { return step_impl(h, *h->f64, dt); }

int steps(NbodyHandle* h, int k) {
    State& s = *h->f64;
    for (int i = 0; i < k; ++i) {
        int rc = step_impl(h, s, s.dt);  // Simulation::step, shared.rs:86-88
        if (rc) return rc;
    }
    return NBODY_OK;
}

int update_forces(NbodyHandle* h) {
    State& s = *h->f64;
    if (h->cfg.method == NBODY_BARNES_HUT && !s.bounds_set) return fail(h, NBODY_ERR_INVALID, "nbody_set_bounds has not been called");
    int rc = exchange(h, s);
    if (rc) return rc;
    return forces(h, s);
}

double elapsed(const NbodyHandle* h) { return h->f64->elapsed; }

int stats(NbodyHandle* h, NbodyStats* out) {
    State& s = *h->f64;
    if (h->cfg.method == NBODY_BRUTE_FORCE) {
        unsigned long long* hv = reinterpret_cast<unsigned long long*>(h->h_poison + 4);   // (pinned scratch)
        HIP_TRY(h, hipMemcpyAsync(hv, s.d.inter, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        h->stats.interactions = *hv;
    }
    if (h->d_counters) {
        HIP_TRY(h, hipMemcpyAsync(h->h_counters, h->d_counters, 2 * NBODY_WALK_COUNTER_SLOTS * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        unsigned long long acc_sum = 0, vis_sum = 0;
        for (unsigned k = 0; k < NBODY_WALK_COUNTER_SLOTS; ++k) { acc_sum += h->h_counters[2 * k]; vis_sum += h->h_counters[2 * k + 1]; }
        h->stats.interactions = acc_sum;
        h->stats.force_kernel_interactions = acc_sum;
        h->stats.node_visits = vis_sum;
    }
    *out = h->stats;
    return NBODY_OK;
}

int reset_stats(NbodyHandle* h) {
    State& s = *h->f64;
    HIP_TRY(h, hipMemsetAsync(s.d.inter, 0, sizeof(unsigned long long), h->stream));
    return NBODY_OK;
}

int energy(NbodyHandle* h, double* kinetic, double* potential) {
    State& s = *h->f64;
    if (s.d.n_seg > 1) return fail(h, NBODY_ERR_INVALID, "nbody_energy on a sharded f64 world is not supported (a rank holds the velocities of its own block only)");
    int rc = sync_count(h, s);
    if (rc) return rc;
    const size_t n = s.n_local;
    const size_t blocks = (n + 255) / 256;
    double ke = 0.0, pe = 0.0;
    if (blocks) {
        if (blocks > s.energy_blocks) {
            if (s.d_energy) (void)hipFree(s.d_energy);
            s.d_energy = nullptr; s.energy_blocks = 0;
            HIP_TRY(h, hipMalloc(&s.d_energy, blocks * 2 * sizeof(double)));
            s.energy_blocks = blocks;
        }
        launch_energy(h->stream, s.d, int(n), s.g_soft * s.g_soft, s.d_energy);
        HIP_TRY(h, hipGetLastError());
        std::vector<double> part(blocks * 2);
        HIP_TRY(h, hipMemcpyAsync(part.data(), s.d_energy, blocks * 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        for (size_t b = 0; b < blocks; ++b) { ke += part[2 * b]; pe += part[2 * b + 1]; }
    }
    if (kinetic) *kinetic = ke;
    if (potential) *potential = -0.5 * s.g * pe;  // every unordered pair was met twice
    return NBODY_OK;
}

int tree_export(NbodyHandle* h, double* com_mass, double* width, int32_t* skip, size_t cap, size_t* n_nodes) {
    State& s = *h->f64;
    const size_t n = s.tree_on_device ? s.dev_nodes : s.tree.n_nodes;
    if (n_nodes) *n_nodes = n;
    if (!com_mass && !width && !skip) return NBODY_OK;
    if (n > cap) return fail(h, NBODY_ERR_CAPACITY, "tree export buffer too small");
    std::vector<nbody::NodeRecT<double>> from_device;
    if (s.tree_on_device) {
        from_device.resize(n);
        if (n) HIP_TRY(h, hipMemcpy(from_device.data(), s.d_nodes, n * sizeof(Node64), hipMemcpyDeviceToHost));
    }
    const nbody::NodeRecT<double>* nodes = s.tree_on_device ? from_device.data() : s.tree.nodes;
    for (size_t i = 0; i < n; ++i) {
        const nbody::NodeRecT<double>& r = nodes[i];
        if (com_mass) { com_mass[4 * i] = r.a.x; com_mass[4 * i + 1] = r.a.y; com_mass[4 * i + 2] = r.a.z; com_mass[4 * i + 3] = r.a.m; }
        if (width) width[i] = std::sqrt(r.b.w2);  // exact: w2 is the rounded square of the width
        if (skip) skip[i] = r.b.skip;
    }
    return NBODY_OK;
}

}  // namespace nbody64
