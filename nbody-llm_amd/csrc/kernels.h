// kernels.h -- host-callable launchers of the gfx950 kernels (internal to libnbody_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

namespace nbody {

// Launch-shape and scheme knobs of ONE handle (NbodyHandle::tune): set at nbody_create from the documented NBODY_*
// environment switches, changed with nbody_set_tuning.  The launchers read the knobs of the handle the calling thread
// is serving (tuning(); every C entry point binds them), so two handles -- or two rank threads of one process -- never
// see each other's settings, and the library exports no mutable tuning globals.
struct Tuning {
    int cross_sym = 1;          // sharded fast math: 1 = every pair between shards once (partial sums travel back), 0 = one-sided  [NBODY_CROSS_SYM]
    int sym_packed = 1;         // 1: packed-fp32 pair evaluation (pair_evals_pk), 0: scalar                                       [NBODY_SYM_PACKED]
    int bf_fast_variant = 0;    // 0: symmetric kernels where they apply, 1..: the LDS-tiled one-sided forms (kernels_bf.hip)       [NBODY_BF_VARIANT]
    int bh_walk_duo = -1;       // Barnes-Hut fast walk: bodies of the tree order per lane sharing their node fetches (k_bh_walk_duo):
                                // -1 from the body count (walk_plan), 0 / 1 one body per lane (k_bh_walk), 2 3 4 6 8                              [NBODY_BH_DUO]
    int bh_walk_xcd = 1;        // k_bh_walk_duo / k_bh_walk_fast64: each XCD walks one eighth of the tree order (its L2 then holds that region's deep nodes)  [NBODY_BH_XCD]
    int let_list_div = 4;       // spatial shards: export / import buffers start at (slice node capacity) / this (they grow when a step needs more)
    int sym_ipt = 0;            // k_bf_sym: bodies per lane of a resident set: 0 from the shard's size (4 up to 10 240 bodies, else 8), 4, 8        [NBODY_SYM_IPT]
    int sym_wpb = 4;            // k_bf_sym: waves per workgroup: 4 (default), 8, 12 or 16                                           [NBODY_SYM_WPB]
    int sym_rounds = 1;         // k_bf_sym: rounds of workgroups per CU
    int sym_k = 0;              // k_bf_sym: waves (slices) per resident set; 0 = by the plan's rule
    int sym_min_bodies = 1024;  // single shard: below this the LDS-tiled one-sided kernel runs instead of k_bf_sym (measured: 18 against 26 us at 1 024 bodies, 28 against 43 at 4 096)
    int sym_reduce_split = 1;   // plane reduction: 1 = several waves per 64 bodies, 0 = one thread per body
    int cross_slots = 3072;     // k_bf_cross: waves the chunk visits are dealt to
    int cross_ipt = 0;          // k_bf_cross: resident bodies per lane: 0 = by rule, 4, 8
    int cross_wpb = 4;          // k_bf_cross: waves per workgroup: 4, 8, 12
    int bh_walk_split = 0;      // node-range segments per body group: 0 = automatic                                                [NBODY_BH_SPLIT]
    int bh_walk_order = 1;      // 1: a group's segments are dispatched nearest-first (heaviest first), 0: in index order
    int bh_reduce_split = 1;    // 1: four waves per 64 bodies in the walk's plane reduction when there are >= 8 segments
    int tree_max_tie = 64;      // device build: largest run of equal 63-bit keys that gets second keys (beyond: "too deep")
    // the following select code that only the tuning build carries (make -C csrc tuning: -DNBODY_TUNING); the release
    // library refuses any value but the default
    int bh_walk_variant = 0;    // 1 wave-cooperative, 2 two lanes per body, 3 hot records in LDS, 4 cooperative window, 5 cooperative block walk  [NBODY_BH_VARIANT]
    int bh_walk_lds_block = 1024;  // variant 3: threads per workgroup                                                              [NBODY_BH_LDS_BLOCK]
    int bh_hot_cap = 2048;      // variant 3: node records staged in LDS per workgroup                                              [NBODY_BH_HOT]
    int bh_walk_debug = 0;      // 1: per-wave start/end stamps (tools/bh_wave_times.py)
    int sym_debug = 0;          // 4: in-kernel cycle stamps (tools/sym_cycles.py); 5-7: timing experiments that do not compute the forces
};
const Tuning& tuning();                 // of the handle this thread is serving; the defaults outside a call
// The fast Barnes-Hut walk's shape for `n_order` bodies (one launch: bodies per lane x node-range segments), from the sweep
// profiles/r03_bh_walk_plan_sweep.txt: the walk is bound by the L1's address rate for divergent gathers, neighbours in tree
// order visit almost the same nodes, so a lane that walks several of them in lockstep fetches their UNION once -- as many
// per lane as still leaves ~32 768 waves' worth of (lane groups x segments) for the chip.
struct WalkPlan { int bodies_per_lane; int segments; };
WalkPlan walk_plan(size_t n_order, bool fast_math, int max_segments, float theta2);
void bind_tuning(const Tuning* t);      // (nullptr: back to the defaults)

// Device-resident body state of one shard.  Positions of ALL shards live in `pos_all`
// (world_size segments of `seg_cap` float4 {x,y,z,m}); velocities and accelerations only for
// the shard's own segment.  Body counts are device-resident so that bodies can leave the box
// (Vec::retain, brute_force.rs:86) without a host round trip.
struct Shard {
    float4* pos_all = nullptr;   // [n_seg * seg_cap]  {x, y, z, mass}
    float4* vel = nullptr;       // [seg_cap]          {vx, vy, vz, 0}
    float4* acc = nullptr;       // [seg_cap]          {ax, ay, az, 0}
    int* seg_count = nullptr;    // [n_seg] bodies alive per segment
    int* escaped = nullptr;      // [1] bodies of the own segment flagged out of bounds by drift
    unsigned char* keep = nullptr;  // [seg_cap] 1 = in bounds
    // K4 (parallel retain): per-tile status words of the decoupled look-back {epoch | flag | count} and the epoch
    unsigned long long* tile_state = nullptr;   // [ceil(seg_cap / 1024)]
    int* epoch = nullptr;                       // [1]
    // Barnes-Hut steps enqueued without a host round trip (device tree): [0] != 0 = "poisoned" (a build needed the
    // host: deeper than the device build's 21 levels, or more nodes than allocated) -- every kernel that changes the
    // state then does nothing until the host has dealt with it; [1] = steps completed since the host last looked
    int* poison = nullptr;
    int* ids = nullptr;                         // [seg_cap] spatial shards: index of each own body in the uploaded vector (moves with it)
    unsigned long long* inter = nullptr;        // [1] brute force: directed interactions evaluated, n_own * (n_total - 1) per force pass from the LIVE counts
    int n_seg = 1;
    int seg_cap = 0;
    int my_seg = 0;
    float4* own_pos() const { return pos_all + size_t(my_seg) * seg_cap; }
    int* own_count() const { return seg_count + my_seg; }
};

struct BoundsF {  // Bounds::min()/max() (shared.rs:223-229) evaluated once on the host in f32
    float lo[3];
    float hi[3];
};

// K0: PointParticle<f32,3> AoS (stride in floats) <-> SoA
void launch_aos_to_soa(hipStream_t s, const float* aos, int stride_f, int n, float4* pos, float4* vel, float4* acc);
void launch_soa_to_aos(hipStream_t s, float* aos, int stride_f, int n, const float4* pos, const float4* vel, const float4* acc);

// K1: integrate_pre_force (shared.rs:135-140) + Bounds::contains flags (shared.rs:210-212)
void launch_drift_half(hipStream_t s, const Shard& sh, int n_upper, float dt, BoundsF b);
// K4: Vec::retain (brute_force.rs:86): in-place order-preserving compaction, no-op unless *escaped
void launch_compact(hipStream_t s, const Shard& sh, int n_upper);
// K3: integrate_after_force (shared.rs:141-148)
void launch_kick_drift(hipStream_t s, const Shard& sh, int n_upper, float dt);

// K2: BruteForceSimulation::update_forces (brute_force.rs:64-82)
void launch_bf_forces_strict(hipStream_t s, const Shard& sh, int n_upper, float g, float g_soft2);
void launch_bf_forces_fast(hipStream_t s, const Shard& sh, int n_upper, float g, float g_soft2);

// K2, symmetric form (every unordered pair once, both bodies updated): kernels_bf_sym.hip
struct SymPlan {
    int ipt = 8;          // resident bodies per lane
    int A = 0;            // resident sets of 64*ipt bodies
    int K = 0;            // waves per resident set
    int wpb = 16;         // waves per workgroup (CU-sized: 12 by default)
    int res_combine = 0;  // 1: K % wpb == 0, resident-side sums of a workgroup are combined in LDS
    int k_res = 0;        // resident-side planes (K, or K / wpb when combined)
    int sym_sets = 0;     // sets met symmetrically = ceil(A/2) - 1
    int n_planes = 0;     // partial-sum planes (own-shard kernels + k_os one-sided planes of a sharded run)
    int k_os = 0;         // slices per resident set of the one-sided remote kernel (0 = single shard)
    size_t n_pad = 0;     // padded body count (A * 64 * ipt)
    size_t plane_stride = 0;  // float4 entries per plane (>= n_pad; sharded runs: also >= the shard capacity)
    std::vector<int> bounds;  // K+1 cut points of a set's chunk sequence
};
SymPlan make_sym_plan(int n_upper);
int sym_bodies_per_lane(size_t n);   // 8, or 4 for small shards (Tuning::sym_ipt)
uint64_t sym_main_pairs(const SymPlan& p, size_t n);  // unordered pairs of real bodies k_bf_sym evaluates
void launch_bf_sym_main(hipStream_t s, const Shard& sh, const SymPlan& p, const int* d_bounds, float4* planes,
                        int n_upper, float g_soft2);
void launch_bf_os(hipStream_t s, const Shard& sh, int A, int K, float4* planes, size_t plane_stride, float g_soft2);
void launch_bf_sym_tail(hipStream_t s, const Shard& sh, const SymPlan& p, float4* planes, int n_upper, float g,
                        float g_soft2, const float* kick_dt);

// K2, symmetric across shards (kernels_bf_cross.hip)
struct CrossPartners {  // passed to the kernels by value
    static constexpr int kMax = 8;
    int n = 0;
    int seg[kMax];          // partner shard
    int c0[kMax], c1[kMax]; // its chunks (of 64 bodies) this GPU evaluates
    int a0[kMax], a1[kMax]; // own resident sets that take part
};
struct CrossPlan {
    static constexpr int kMaxPlanes = 256;
    int ipt = 8;               // resident bodies per lane (8, or 4 for small shards)
    int A = 0;                 // own resident sets of 64*ipt bodies
    CrossPartners parts;       // shards this GPU is resident for
    int n_recv = 0;            // shards that send partial sums for the own bodies ...
    int recv_from[CrossPartners::kMax];  // ... in this (rank-distance) order
    int k_res = 0;             // resident-side planes
    std::vector<int4> slices;  // {set, first chunk, last chunk (of the set's partner-chunk sequence), plane}
};
CrossPlan make_cross_plan(int rank, int world, int seg_cap, int n_own_upper);
void launch_bf_cross(hipStream_t s, const Shard& sh, const CrossPlan& p, const int4* d_slices, float4* res_planes,
                     float4* xplanes, float4* send, size_t plane_stride, float g_soft2);

// K5: BarnesHutSimulation::calc_force (barnes_hut.rs:185-203) over a linearised octree
struct TreeDev {
    const float4* nodes = nullptr;   // 2 per node: {com.x, com.y, com.z, mass}, {width^2, skip (int bits), hot score (int bits), leaf body}
    int n_nodes = 0;
    const int* order = nullptr;      // own bodies (index into the own segment) in tree order
    int n_order = 0;
    // node-range split of the walk (kernels_bh.hip WalkSplit); n_split = 1: none
    int n_split = 1;
    const int* split_first = nullptr;   // [n_split + 1]
    const int* split_anc = nullptr;     // [n_split][192]
    const int* split_n_anc = nullptr;   // [n_split]
    float4* split_planes = nullptr;     // [n_split][split_stride]
    size_t split_stride = 0;
    // strict math, reference leaf rule: per-lane stack of open cells {partial sum, end index} for the
    // reference's nested sums (k_bh_walk_nested); [NBODY_MAX_TREE_DEPTH + 1][nested_stride]
    float4* nested_stack = nullptr;
    size_t nested_stride = 0;
    // fast math: the walk with the most-visited node records staged in LDS (k_bh_walk_lds); hot_cap = 0: off
    float4* walk = nullptr;          // [n_nodes + 1] records with explicit links {w^2, skip link, open link, pre-order index}
    float4* hot = nullptr;           // [hot_cap] the records every workgroup copies into its LDS
    int* unified = nullptr;          // [n_nodes + 1] pre-order index -> link value (< hot_cap: LDS slot, else index + hot_cap)
    int* hot_info = nullptr;         // [0] slot counter of the pass under way, [1] nodes the last pass flagged
    int hot_cap = 0;                 // LDS table entries
    int hot_threshold = 0;           // a node is staged if NodeB::hot >= this
    // unsynchronised steps (device tree, no read-back): kernels stop when *poison != 0 and take the number of bodies
    // to walk from the device
    const int* poison = nullptr;
    const int* n_order_dev = nullptr;
    int store_work = 0;              // the plain walk leaves every body's visit count in acc.w (spatial shards)
    // fast math: the cooperative block walk (k_bh_walk_block) reads this level-order copy of `nodes`, built per step
    float4* bfs = nullptr;           // [n_nodes] records {com, mass | w^2, pre-order index, pre-order skip, first child | last flag}
    void* bfs_ws = nullptr;          // workspace of build_bfs_layout
    size_t bfs_cap = 0;              // nodes both are sized for
};
// device-side octree build (kernels_tree.hip)
struct TreeDevWork {  // arrays of the last build (inside its workspace)
    const unsigned long long* keys = nullptr;   // sorted keys
    const signed char* delta = nullptr;         // levels shared by neighbouring sorted bodies
    const int* base = nullptr;                  // first node of every sorted body
    const int* ids = nullptr;                   // sorted position -> body
    const unsigned long long* keys2 = nullptr;  // levels 21..41, defined inside groups of equal keys only
    const int* wpre = nullptr;                  // exclusive prefix sums of the bodies' weights over the sorted order (tree_scan_sorted with weights)
    const void* incl = nullptr;                 // inclusive f64 prefix sums {m, m x, m y, m z} over the sorted bodies (4 doubles each)
};
struct TreeCat {  // sharded runs: side buffer of the device build
    float4* pos = nullptr;     // live bodies of all segments, concatenated in segment order
    int* flags = nullptr;
    int* base = nullptr;
    int* own_order = nullptr;  // own bodies in tree order, indices into the own segment
    int* info = nullptr;       // [0] total bodies, [1] first own body in the concatenation, [2] own count
};
size_t tree_build_workspace_bytes(size_t n_cap);
size_t tree_build_tmp_bytes(size_t n_cap);
size_t tree_cat_bytes(size_t n_cap);
TreeCat tree_cat_layout(void* buf, size_t n_cap);
void launch_tree_cat(hipStream_t s, const Shard& sh, const TreeCat& c);
// the same for F = f64 (double4 positions; the TreeCat's pos is unused)
size_t tree_cat_bytes64(size_t n_cap);
TreeCat tree_cat_layout64(void* buf, size_t n_cap, double4** pos_cat);
void launch_tree_cat64(hipStream_t s, const double4* pos_all, const int* seg_count, int n_seg, int seg_cap, int my_seg, double4* pos_cat, int* info);
int launch_tree_own_order(hipStream_t s, const int* order, const TreeCat& c, int n_total_upper, void* tmp, size_t tmp_bytes);
// The walk's node-range split points ride in the emit's launch as extra workgroups when the caller asks for them here
// (unsynchronised single-shard steps: one launch of ~8 us less per step); n_split = 0: not wanted.
struct TreeSplitReq { int n_split = 0; int* first = nullptr; int* n_anc = nullptr; int* anc = nullptr; int max_anc = 0; const int* info = nullptr; int* poison = nullptr; };
int build_octree_device(hipStream_t s, const float4* pos, const int* d_count, int n_upper, const float center[3],
                        float width, void* workspace, size_t n_cap, float4* nodes, int node_cap, int* order,
                        int* out_info, TreeDevWork* work, int want_hot = 0 /* fill NodeB::hot (LDS-staged walk only) */, const TreeSplitReq* split = nullptr);
// level-order copy of a pre-order node array for the cooperative block walk
size_t bfs_workspace_bytes(size_t n_cap);
int build_bfs_layout(hipStream_t s, const float4* nodes, int n_nodes, void* workspace, size_t n_cap, float4* out);
// F = f64 (nbody_f64.cpp): double4 bodies, Node64 records (kernels_f64.h)
}  // namespace nbody
namespace nbody64 { struct Node64; }
namespace nbody {
int build_octree_device_f64(hipStream_t s, const double4* pos, const int* d_count, int n_upper, const double center[3], double width,
                            void* workspace, size_t n_cap, nbody64::Node64* nodes, int node_cap, int* order, int* out_info,
                            TreeDevWork* work, const TreeSplitReq* split = nullptr);
// the build in two halves (spatial shards need the sorted keys of all ranks' ends before the second one)
int tree_sort_keys(hipStream_t s, const float4* pos, const int* d_count, int n_upper, const float center[3], float width,
                   void* workspace, size_t n_cap, int* out_info, TreeDevWork* work);
int tree_emit_sorted(hipStream_t s, const float4* pos, const int* d_count, int n_upper, float width, void* workspace, size_t n_cap,
                     float4* nodes, int node_cap, int* order, int* out_info, int want_hot, const int* edge, const TreeSplitReq* split = nullptr);
// the two halves of tree_emit_sorted, for the spatial-shard build (the nodes are emitted after an exchange, at an offset
// given by *node_offset, with each node's parent and depth)
int tree_scan_sorted(hipStream_t s, const float4* pos, const int* d_count, int n_upper, void* workspace, size_t n_cap, int* out_info,
                     const int* edge, const float4* weight_src = nullptr /* acc: .w = last walk's visit count */);
// the weight a body's visit count stands for in the balance of spatial shards (>= 1; scaled so that 2^31 is far away)
__host__ __device__ inline int body_weight(float visits) { const int v = int(visits) >> 4; return v < 1 ? 1 : v; }
int tree_emit_nodes(hipStream_t s, const float4* pos, const int* d_count, int n_upper, float width, void* workspace, size_t n_cap,
                    float4* nodes, int node_cap, int slice_cap, int* order, int* out_info, int want_hot, const int* edge,
                    const int* node_offset, int* parent, unsigned char* depth);
void launch_tree_split_anc(hipStream_t s, const TreeDevWork& work, int n, int n_nodes, int n_split, int* first,
                           int* n_anc, int* anc, int max_anc, const int* info = nullptr /* device: {n_nodes, flags, n}: overrides n, n_nodes */,
                           int* poison = nullptr /* made sticky when the build raised a flag */);

// the walk's {accepted, visited} counters: this many u64 pairs, to be summed by the reader
#define NBODY_WALK_COUNTER_SLOTS 1024u
// split points and their ancestors from the node array itself (first [n_split + 1], n_anc [n_split], anc [n_split][192]; range: device {begin, end})
void launch_walk_split_scan(hipStream_t s, const float4* nodes, const int* range, int n_split, int* first, int* n_anc, int* anc, int first_given = 0);
void launch_bh_walk(hipStream_t s, const Shard& sh, const TreeDev& t, float g, float g_soft2, float theta2,
                    int fast_math, unsigned long long* counters /* [NBODY_WALK_COUNTER_SLOTS][2]: accepted, visited */, int leaf_direct = 0,
                    const float* kick_dt = nullptr /* fuse integrate_after_force into the plane reduction */,
                    int* kicked = nullptr /* out: 1 if it was applied */);

// diagnostics: f64 energies of the own segment against all segments; out = {KE, PE_pairs_sum}
void launch_energy(hipStream_t s, const Shard& sh, int n_upper, double g_soft2, double* out2);

}  // namespace nbody
