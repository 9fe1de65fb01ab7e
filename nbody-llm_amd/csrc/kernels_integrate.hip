// kernels_integrate.hip -- O(N) kernels around the force pass (gfx950).
//
// K0  AoS <-> SoA transposition of PointParticle<f32,3> records (shared.rs:151-158)
// K1  drift_half    = LeapFrogIntegrator::integrate_pre_force (shared.rs:135-140)
//                     + the Bounds::contains test (shared.rs:210-212) that retain() applies next
// K4  compact       = Vec::retain (brute_force.rs:86, barnes_hut.rs:267), order preserving, one pass over many workgroups
// K3  kick_drift    = LeapFrogIntegrator::integrate_after_force (shared.rs:141-148)
//
// All are pure streaming kernels, 16 B per lane per access (1 KiB per wave instruction).
// Compiled with -ffp-contract=off: (v*0.5)*dt and a*dt are rounded products, then added, exactly
// as the reference's nalgebra expressions evaluate.
#include "kernels.h"

#include <algorithm>

namespace nbody {

__global__ __launch_bounds__(256) void k_aos_to_soa(const float* __restrict__ aos, int stride_f, int n,
                                                    float4* __restrict__ pos, float4* __restrict__ vel,
                                                    float4* __restrict__ acc) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* p = aos + size_t(k) * stride_f;
    pos[k] = make_float4(p[0], p[1], p[2], p[9]);
    if (vel) vel[k] = make_float4(p[3], p[4], p[5], 0.f);  // other shards' segments carry positions only
    if (acc) acc[k] = make_float4(p[6], p[7], p[8], 0.f);
}

__global__ __launch_bounds__(256) void k_soa_to_aos(float* __restrict__ aos, int stride_f, int n,
                                                    const float4* __restrict__ pos, const float4* __restrict__ vel,
                                                    const float4* __restrict__ acc) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    float4 p = pos[k], v = vel[k], a = acc[k];
    float* o = aos + size_t(k) * stride_f;
    o[0] = p.x; o[1] = p.y; o[2] = p.z;
    o[3] = v.x; o[4] = v.y; o[5] = v.z;
    o[6] = a.x; o[7] = a.y; o[8] = a.z;
    o[9] = p.w;
}

__global__ __launch_bounds__(256) void k_drift_half(float4* __restrict__ pos, const float4* __restrict__ vel,
                                                    const int* __restrict__ count, unsigned char* __restrict__ keep,
                                                    int* __restrict__ escaped, float dt, BoundsF b,
                                                    const int* __restrict__ poison) {
    if (poison && *poison) return;
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *count) return;
    float4 p = pos[k];
    float4 v = vel[k];
    p.x += (v.x * 0.5f) * dt;
    p.y += (v.y * 0.5f) * dt;
    p.z += (v.z * 0.5f) * dt;
    pos[k] = p;
    // inclusive, component-wise; a NaN fails every comparison and is dropped
    bool in = (p.x >= b.lo[0]) && (p.x <= b.hi[0]) && (p.y >= b.lo[1]) && (p.y <= b.hi[1]) &&
              (p.z >= b.lo[2]) && (p.z <= b.hi[2]);
    keep[k] = in ? 1 : 0;
    if (!in) atomicAdd(escaped, 1);
}

__global__ __launch_bounds__(256) void k_kick_drift(float4* __restrict__ pos, float4* __restrict__ vel,
                                                    const float4* __restrict__ acc, const int* __restrict__ count,
                                                    float dt, int* __restrict__ poison) {
    if (poison && *poison) return;
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0 && poison) atomicAdd(poison + 1, 1);   // a step of an unsynchronised Barnes-Hut run is complete
    if (k >= *count) return;
    float4 p = pos[k], v = vel[k], a = acc[k];
    v.x += a.x * dt;
    v.y += a.y * dt;
    v.z += a.z * dt;
    p.x += (v.x * 0.5f) * dt;
    p.y += (v.y * 0.5f) * dt;
    p.z += (v.z * 0.5f) * dt;
    vel[k] = v;
    pos[k] = p;
}

// K4: Vec::retain (brute_force.rs:86, barnes_hut.rs:267) -- order-preserving compaction of the own segment, in
// place, in ONE pass over many workgroups (round 1 walked the segment with a single 1 024-thread workgroup: an escape
// at N = 2^22 serialised 4 096 chunk iterations on one CU, ms-scale).  Launched every step, every workgroup returns at
// once unless drift_half flagged an escape.
//   * a tile = 1 024 consecutive bodies, one per thread; the thread loads its record, the workgroup counts and ranks
//     its survivors (wave ballots + 16 wave totals in LDS);
//   * tiles learn how many survivors precede them by decoupled look-back: a tile publishes {its own count}, then adds
//     up its predecessors' published counts backwards until it meets one that already knows its inclusive prefix, and
//     publishes its own inclusive prefix.  Status words carry the launch's epoch, so they never need resetting;
//   * in place: a survivor moves to an index <= its own, i.e. into the source range of its own or an EARLIER tile.  A
//     tile publishes only after its own records are in registers, and a tile's prefix is built from published words
//     only, so -- by induction over the tiles it looked back over -- every earlier tile has finished reading before
//     this tile knows where to write; inside a tile a barrier separates the loads from the stores.
constexpr int kCompactTile = 1024;
constexpr unsigned long long kTileAgg = 1ull, kTilePrefix = 2ull;
__device__ __forceinline__ unsigned long long tile_word(int epoch, unsigned long long flag, int value) {
    return ((unsigned long long)(unsigned)epoch << 34) | (flag << 32) | (unsigned long long)(unsigned)value;
}

__global__ __launch_bounds__(kCompactTile) void k_compact(float4* __restrict__ pos, float4* __restrict__ vel,
                                                          float4* __restrict__ acc, const unsigned char* __restrict__ keep,
                                                          int* __restrict__ count, int* __restrict__ escaped,
                                                          unsigned long long* __restrict__ tile_state, int* __restrict__ epoch_p,
                                                          const int* __restrict__ poison, int* __restrict__ ids) {
    if (poison && *poison) return;
    if (*escaped == 0) return;
    __shared__ int wave_total[16];
    __shared__ int excl_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int n = *count;
    const int epoch = *epoch_p & 0x3fffffff;
    const int k = tile * kCompactTile + tid;
    const bool kp = (k < n) && keep[k];
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), v = p, a = p;
    int id = 0;
    if (kp) { p = pos[k]; v = vel[k]; a = acc[k]; if (ids) id = ids[k]; }
    const unsigned long long m = __ballot(kp);
    const int in_wave = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_total[wave] = __popcll(m);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's records are in registers ...
    __syncthreads();                                   // ... and so are the whole tile's
    int before = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        const int t = wave_total[w];
        if (w < wave) before += t;
        total += t;
    }
    if (tid == 0) {
        volatile unsigned long long* st = tile_state;
        int excl = 0;
        if (tile == 0) {
            st[0] = tile_word(epoch, kTilePrefix, total);
        } else {
            st[tile] = tile_word(epoch, kTileAgg, total);
            __threadfence();
            for (int j = tile - 1; j >= 0;) {
                const unsigned long long wd = st[j];
                if (int(wd >> 34) != epoch) continue;            // not published in this launch yet: look again
                excl += int(unsigned(wd & 0xFFFFFFFFull));
                if (((wd >> 32) & 3ull) == kTilePrefix) break;   // everything before j is in this word
                --j;
            }
            st[tile] = tile_word(epoch, kTilePrefix, excl + total);
        }
        __threadfence();
        excl_s = excl;
        if (tile == int(gridDim.x) - 1) {   // the last tile's inclusive prefix is the new body count
            *count = excl + total;
            *escaped = 0;
            *epoch_p = (epoch + 1) & 0x3fffffff;
        }
    }
    __syncthreads();
    if (kp) {
        const int d = excl_s + before + in_wave;
        pos[d] = p; vel[d] = v; acc[d] = a;
        if (ids) ids[d] = id;
    }
}

static inline int blocks_for(int n, int bs) { return n <= 0 ? 0 : (n + bs - 1) / bs; }

void launch_aos_to_soa(hipStream_t s, const float* aos, int stride_f, int n, float4* pos, float4* vel, float4* acc) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_aos_to_soa, dim3(blocks_for(n, 256)), dim3(256), 0, s, aos, stride_f, n, pos, vel, acc);
}
void launch_soa_to_aos(hipStream_t s, float* aos, int stride_f, int n, const float4* pos, const float4* vel, const float4* acc) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_soa_to_aos, dim3(blocks_for(n, 256)), dim3(256), 0, s, aos, stride_f, n, pos, vel, acc);
}
void launch_drift_half(hipStream_t s, const Shard& sh, int n_upper, float dt, BoundsF b) {
    if (n_upper <= 0) return;
    hipLaunchKernelGGL(k_drift_half, dim3(blocks_for(n_upper, 256)), dim3(256), 0, s, sh.own_pos(), sh.vel,
                       sh.own_count(), sh.keep, sh.escaped, dt, b, sh.poison);
}
void launch_compact(hipStream_t s, const Shard& sh, int n_upper) {
    if (n_upper <= 0) return;
    hipLaunchKernelGGL(k_compact, dim3(blocks_for(n_upper, kCompactTile)), dim3(kCompactTile), 0, s, sh.own_pos(), sh.vel, sh.acc,
                       sh.keep, sh.own_count(), sh.escaped, sh.tile_state, sh.epoch, sh.poison, sh.ids);
}
void launch_kick_drift(hipStream_t s, const Shard& sh, int n_upper, float dt) {
    // (launched even for an empty shard: the step counter of an unsynchronised run rides in it)
    hipLaunchKernelGGL(k_kick_drift, dim3(std::max(1, blocks_for(n_upper, 256))), dim3(256), 0, s, sh.own_pos(), sh.vel, sh.acc,
                       sh.own_count(), dt, sh.poison);
}

}  // namespace nbody
