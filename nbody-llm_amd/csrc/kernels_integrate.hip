// kernels_integrate.hip -- O(N) kernels around the force pass (gfx950).
//
// K0  AoS <-> SoA transposition of PointParticle<f32,3> records (shared.rs:151-158)
// K1  drift_half    = LeapFrogIntegrator::integrate_pre_force (shared.rs:135-140)
//                     + the Bounds::contains test (shared.rs:210-212) that retain() applies next
// K4  compact       = Vec::retain (brute_force.rs:86, barnes_hut.rs:267), order preserving
// K3  kick_drift    = LeapFrogIntegrator::integrate_after_force (shared.rs:141-148)
//
// All are pure streaming kernels, 16 B per lane per access (1 KiB per wave instruction).
// Compiled with -ffp-contract=off: (v*0.5)*dt and a*dt are rounded products, then added, exactly
// as the reference's nalgebra expressions evaluate.
#include "kernels.h"

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
                                                    int* __restrict__ escaped, float dt, BoundsF b) {
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
                                                    float dt) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
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

// One 1024-thread workgroup walks the segment in ascending chunks.  Within a chunk every thread
// reads its record, the workgroup scans the keep flags (wave ballots + 16 wave totals in LDS),
// and only after a barrier are the survivors written at base + rank.  Destination indices never
// exceed source indices and later chunks are read after earlier chunks are written, so the
// compaction is safe in place and preserves order like Vec::retain.  Launched every step, it
// returns at once unless drift_half flagged an escape.
__global__ __launch_bounds__(1024) void k_compact(float4* __restrict__ pos, float4* __restrict__ vel,
                                                  float4* __restrict__ acc, const unsigned char* __restrict__ keep,
                                                  int* __restrict__ count, int* __restrict__ escaped) {
    if (*escaped == 0) return;
    __shared__ int wave_total[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = *count;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int c = 0; c < n; c += 1024) {
        int k = c + tid;
        bool kp = (k < n) && keep[k];
        float4 p, v, a;
        if (kp) { p = pos[k]; v = vel[k]; a = acc[k]; }
        unsigned long long m = __ballot(kp);
        int in_wave = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_total[wave] = __popcll(m);
        __syncthreads();  // all reads of this chunk are done; wave totals visible
        int before = 0, total = 0;
        for (int w = 0; w < 16; ++w) {
            int t = wave_total[w];
            if (w < wave) before += t;
            total += t;
        }
        int base = base_s;
        if (kp) {
            int d = base + before + in_wave;
            pos[d] = p; vel[d] = v; acc[d] = a;
        }
        __syncthreads();  // everyone has read base_s / wave_total
        if (tid == 0) base_s = base + total;
        __syncthreads();
    }
    if (tid == 0) {
        *count = base_s;
        *escaped = 0;
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
                       sh.own_count(), sh.keep, sh.escaped, dt, b);
}
void launch_compact(hipStream_t s, const Shard& sh) {
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, s, sh.own_pos(), sh.vel, sh.acc, sh.keep, sh.own_count(),
                       sh.escaped);
}
void launch_kick_drift(hipStream_t s, const Shard& sh, int n_upper, float dt) {
    if (n_upper <= 0) return;
    hipLaunchKernelGGL(k_kick_drift, dim3(blocks_for(n_upper, 256)), dim3(256), 0, s, sh.own_pos(), sh.vel, sh.acc,
                       sh.own_count(), dt);
}

}  // namespace nbody
