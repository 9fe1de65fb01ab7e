// kernels_f64.h -- launchers of the F = f64 kernels (internal to libnbody_hip.so); see kernels_f64.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

#ifndef NBODY_WALK_COUNTER_SLOTS
#define NBODY_WALK_COUNTER_SLOTS 1024u
#endif

namespace nbody64 {

// device-resident state of an f64 handle: the own shard, and (index-block shards) the positions
// and live counts of every shard, refreshed by the per-step exchange -- pos / count point at the own segment of those
struct Dev {
    double4* pos_all = nullptr;   // [n_seg][cap]
    int* seg_count = nullptr;     // [n_seg]
    int n_seg = 1, my_seg = 0;
    double4* pos = nullptr;   // [cap] {x, y, z, mass}: = pos_all + my_seg * cap
    double4* vel = nullptr;   // [cap]
    double4* acc = nullptr;   // [cap]
    int* count = nullptr;     // live bodies of the own shard: = seg_count + my_seg
    int* escaped = nullptr;   // [1]
    unsigned char* keep = nullptr;            // [cap]
    unsigned long long* tile_state = nullptr; // [ceil(cap / 1024) + 1]
    int* epoch = nullptr;                     // [1]
    unsigned long long* inter = nullptr;      // [1] brute force: directed interactions, from the live count
    int cap = 0;
};

struct Bounds64 { double lo[3]; double hi[3]; };   // Bounds::min()/max() (shared.rs:223-229) evaluated on the host in f64

// one node of the linearised octree: NodeRecT<double> of octree_host.h (64 bytes)
struct alignas(64) Node64 {
    double x, y, z, m;   // centre of mass, mass
    double w2;           // width^2
    int skip, hot, body, pad;
    double pad2;
};
static_assert(sizeof(Node64) == 64, "Node64 layout");

struct alignas(32) Open64 { double x, y, z; int end; int pad; };   // an open cell of the nested walk: its partial sum and where it ends

void launch_aos_to_soa(hipStream_t s, const double* aos, int stride_d, int n, const Dev& d, size_t first);
void launch_soa_to_aos(hipStream_t s, double* aos, int stride_d, int n, const Dev& d);
void launch_aos_to_pos(hipStream_t s, const double* aos, int stride_d, int n, double4* pos);   // another shard's block: positions only
void launch_drift_half(hipStream_t s, const Dev& d, int n_upper, double dt, const Bounds64& b);
void launch_compact(hipStream_t s, const Dev& d, int n_upper);
void launch_kick_drift(hipStream_t s, const Dev& d, int n_upper, double dt);
void launch_bf_strict(hipStream_t s, const Dev& d, int n_upper, double g, double eps2);
void launch_bh_walk(hipStream_t s, const Dev& d, const Node64* nodes, int n_nodes, const int* order, int n_order, double g, double eps2,
                    double theta2, unsigned long long* counters, int leaf_direct, Open64* stack, size_t stack_stride);
// fast f64 walk (NBODY_MATH_FAST on an f64 handle): one running sum per lane instead of the reference's nested sums,
// rsqrt + FMA instead of sqrt and divide, and the node index range cut into n_seg segments walked by different waves
// (kernels_bh.hip WalkSplit: a body's walk enters segment k where the replay of the opening tests of first[k]'s
// ancestors says it would); the K partial sums are added in segment order by launch_bh_reduce64.
struct WalkSplit64 {
    int n_seg;
    const int* first;      // [n_seg + 1]
    const int* anc;        // [n_seg][kMaxAnc64]
    const int* n_anc;      // [n_seg]
    double4* planes;       // [n_seg][plane_stride], by tree-order position
    size_t plane_stride;
};
constexpr int kMaxAnc64 = 192;
void launch_bh_walk_fast(hipStream_t s, const Dev& d, const Node64* nodes, int n_nodes, const int* order, int n_order, double g, double eps2,
                         double theta2, unsigned long long* counters, int leaf_direct, const WalkSplit64& split, int bodies_per_lane = 1,
                         const double* kick_dt = nullptr /* fuse integrate_after_force into the plane reduction */, int* kicked = nullptr);
void launch_energy(hipStream_t s, const Dev& d, int n_upper, double eps2, double* out2);

}  // namespace nbody64
