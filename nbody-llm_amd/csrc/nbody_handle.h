// nbody_handle.h -- the handle behind include/nbody_hip.h (internal to libnbody_hip.so).
#pragma once
#include "../../include/nbody_hip.h"
#include "kernels.h"
#include "octree_host.h"

#include "transport.h"

#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <utility>
#include <vector>

using nbody::BoundsF;
using nbody::Shard;

namespace nbody64 { struct State; }   // F = f64 handles (nbody_f64.cpp)
namespace nbody { namespace let { struct State; } }   // spatial shards (nbody_let.cpp)

struct NbodyHandle {
    NbodyConfig cfg{};
    nbody::Tuning tune;        // this handle's launch-shape and scheme knobs (nbody_set_tuning; NBODY_* environment at create)
    int device = 0;
    hipStream_t stream = nullptr;
    Shard sh;
    float g = 1.0f, g_soft = 0.0f, dt = 1e-3f, theta2 = 0.5f;  // shared.rs:69-78
    float center[3] = {0.f, 0.f, 0.f};
    float width = 0.f;
    BoundsF bnd{};
    bool bounds_set = false;
    float elapsed = 0.f;

    size_t n_local = 0;        // host view of the own body count (an upper bound while count_dirty)
    bool count_dirty = false;  // drift may have dropped bodies since n_local was read
    std::vector<int> seg_count_host;  // host view of every segment's count (upper bounds likewise)
    size_t first_global = 0, n_at_upload = 0;

    float* d_aos = nullptr;    // device staging for PointParticle records
    float* h_aos = nullptr;    // pinned host staging
    size_t aos_cap = 0;        // records

    // Barnes-Hut
    std::unique_ptr<nbody::WorkerPool> pool;
    nbody::HostTree tree;
    nbody::BuildScratch tree_scratch;
    float4* d_nodes = nullptr;  // 2 float4 per node: {com, mass}, {width^2, skip, width, leaf body}
    int* d_order = nullptr;
    size_t d_node_cap = 0, d_order_cap = 0;
    float* h_pos = nullptr;    // pinned: all segments' positions
    int* h_counts = nullptr;   // pinned: all segments' counts
    std::vector<int32_t> own_order;
    void* d_tree_ws = nullptr;   // device-build workspace (keys, sort buffers, scans)
    void* d_tree_cat = nullptr;  // sharded device build: concatenated positions, own-order list
    float4* d_nested_stack = nullptr;  // strict Barnes-Hut: per-lane stack of open cells (k_bh_walk_nested)
    size_t nested_cap = 0;
    int nested_levels = 0;
    size_t tree_ws_cap = 0;      // bodies it is sized for
    int* d_tree_info = nullptr;  // [3] node count, flags, bodies in the tree
    int* h_tree_info = nullptr;  // pinned
    bool tree_on_device = false; // the last tree was built on the device (export copies it back)
    int* d_split = nullptr;      // [33 + 32 + 32*192] ints: first[], n_anc[], anc[][192]
    int* h_split = nullptr;      // pinned mirror
    float4* d_walk_planes = nullptr;
    size_t walk_planes_cap = 0;  // float4 entries
    // fast walk with the most-visited records in LDS (kernels_bh.hip, variant 3)
    float4* d_walk = nullptr;    // [walk_cap + 1] records with explicit links
    int* d_unified = nullptr;    // [walk_cap + 1]
    size_t walk_cap = 0;         // nodes
    float4* d_bfs = nullptr;     // cooperative block walk (variant 5): level-order copy of the nodes
    void* d_bfs_ws = nullptr;
    size_t bfs_cap = 0;          // nodes
    float4* d_hot = nullptr;     // [hot_cap] records
    int hot_cap = 0;
    int* d_hot_info = nullptr;   // [2] slot counter, nodes flagged by the last pass
    int* h_hot_info = nullptr;   // pinned; refreshed after every walk, read after the next step's first sync
    int hot_threshold = 0;       // NodeB::hot >= this -> staged; steered so that ~hot_cap nodes qualify
    size_t hot_threshold_n = 0;  // body count the threshold was initialised for
    unsigned long long* d_counters = nullptr;  // [NBODY_WALK_COUNTER_SLOTS][2] accepted, visited (summed on read)
    unsigned long long* h_counters = nullptr;  // pinned

    // symmetric all-pairs kernel (fast math; single shard: n >= Tuning::sym_min_bodies)
    nbody::SymPlan sym_plan;
    int* d_sym_bounds = nullptr;
    float4* d_planes = nullptr;
    size_t planes_cap = 0;  // float4 entries
    int sym_waves = 0;
    // symmetric scheme across shards (kernels_bf_cross.hip)
    nbody::CrossPlan cross;
    bool cross_on = false;
    int4* d_cross_slices = nullptr;
    size_t cross_slices_cap = 0;
    float4* d_xplanes = nullptr;   // [parts.n][A][plane_stride] travelling-side sums for other shards' bodies
    float4* d_send = nullptr;      // [parts.n][plane_stride] what goes back to their owners
    size_t xplanes_cap = 0, send_cap = 0;
    int recv_plane0 = 0;           // first plane that receives the other shards' partial sums
    bool tail_pending = false;     // the plane reduction has not been launched yet (waits for the partials)
    bool partials_in_flight = false;
    hipEvent_t ev_partials_ready = nullptr, ev_partials_done = nullptr;
    bool kick_pending = false;  // step_end asks the force pass to fuse integrate_after_force if it can
    float kick_dt = 0.f;
    uint64_t sym_pairs = 0;   // unordered pairs the rotation kernel covers at the current n_local
    size_t sym_pairs_n = 0;

    // Barnes-Hut with the device build, single shard: steps are enqueued without reading anything back.  A build
    // that needs the host (deeper than 21 levels, node array too small) sets a sticky flag on the device that turns
    // every later state-changing kernel into a no-op; the host looks at it at the next synchronisation point and
    // replays from the step that failed (resolve_async).
    bool async_bh = false;
    struct PendingStep { float dt; float elapsed_before; };
    std::vector<PendingStep> pending;   // steps enqueued since the host last confirmed the device's progress
    bool last_step_async = false;
    bool host_tree_once = false;        // the next force pass builds its tree on the host (the replayed step)
    int* d_poison = nullptr;            // [2] sticky flags, steps completed (Shard::poison)
    int* h_poison = nullptr;            // pinned [8]: poison[2] + tree info[3]

    // diagnostics
    NbodyStats stats{};
    bool profiling = false;
    int profile_every = 1;       // bracket every k-th force-kernel launch with events (nbody_set_profiling(h, k))
    unsigned profile_tick = 0;
    bool timed_this = false;     // the launch under way is one of the bracketed ones
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending, ev_free;
    double* d_energy = nullptr;
    size_t energy_blocks = 0;

    // multi-GPU: what carries the exchanges (RCCL, or the one-device transport of transport_ipc.hip)
    std::unique_ptr<nbody::Transport> tp;
    bool comm_ready = false;
    hipStream_t comm_stream = nullptr;   // the exchange runs here, beside the own-shard force kernel
    hipEvent_t ev_drifted = nullptr, ev_gathered = nullptr;
    bool exchange_in_flight = false;

    nbody64::State* f64 = nullptr;   // NbodyConfig.dtype == NBODY_F64: the whole state lives here
    nbody::let::State* let = nullptr; // NbodyConfig.shard_mode == NBODY_SHARD_SPATIAL: the halo-exchange machinery

    std::string err;
};

