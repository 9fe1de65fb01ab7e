// kernels_f64.h -- launchers of the F = f64 kernels (internal to libnbody_hip.so); see kernels_f64.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

#ifndef NBODY_WALK_COUNTER_SLOTS
#define NBODY_WALK_COUNTER_SLOTS 1024u
#endif

namespace nbody64 {

// device-resident state of an f64 handle (one shard)
struct Dev {
    double4* pos = nullptr;   // [cap] {x, y, z, mass}
    double4* vel = nullptr;   // [cap]
    double4* acc = nullptr;   // [cap]
    int* count = nullptr;     // [1] live bodies
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
void launch_drift_half(hipStream_t s, const Dev& d, int n_upper, double dt, const Bounds64& b);
void launch_compact(hipStream_t s, const Dev& d, int n_upper);
void launch_kick_drift(hipStream_t s, const Dev& d, int n_upper, double dt);
void launch_bf_strict(hipStream_t s, const Dev& d, int n_upper, double g, double eps2);
void launch_bh_walk(hipStream_t s, const Dev& d, const Node64* nodes, int n_nodes, const int* order, int n_order, double g, double eps2,
                    double theta2, unsigned long long* counters, int leaf_direct, Open64* stack, size_t stack_stride);
void launch_energy(hipStream_t s, const Dev& d, int n_upper, double eps2, double* out2);

}  // namespace nbody64
