// kernels_f64.hip -- the path with F = f64 (the reference's trait is generic over Float, shared.rs:12-44, and its own
// driver instantiates f64: src/main.rs:52-105).  Strict arithmetic only: the reference's rounding sequence in double
// precision, bit for bit against the oracle's f64 instantiation and the committed *_f64_* golden vectors.  One shard;
// the kernels are the f32 strict kernels' shapes (kernels_integrate.hip, kernels_bf.hip k_bf_strict, kernels_bh.hip
// k_bh_walk_nested / k_bh_walk<false, DIRECT>) with double4 state and 64-byte node records.
// Compiled with -ffp-contract=off; f64 sqrt and divide are correctly rounded on gfx950.
#include "kernels_f64.h"
#include "kernels.h"   // nbody::tuning()

#include <algorithm>

namespace nbody64 {

namespace {

// ---- K0: PointParticle<f64,3> AoS (10 doubles, 80 B) <-> SoA
__global__ __launch_bounds__(256) void k_aos_to_soa(const double* __restrict__ aos, int stride_d, int n, double4* __restrict__ pos,
                                                    double4* __restrict__ vel, double4* __restrict__ acc) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const double* p = aos + size_t(k) * stride_d;
    pos[k] = make_double4(p[0], p[1], p[2], p[9]);
    vel[k] = make_double4(p[3], p[4], p[5], 0.0);
    acc[k] = make_double4(p[6], p[7], p[8], 0.0);
}

__global__ __launch_bounds__(256) void k_aos_to_pos(const double* __restrict__ aos, int stride_d, int n, double4* __restrict__ pos) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const double* p = aos + size_t(k) * stride_d;
    pos[k] = make_double4(p[0], p[1], p[2], p[9]);
}

__global__ __launch_bounds__(256) void k_soa_to_aos(double* __restrict__ aos, int stride_d, int n, const double4* __restrict__ pos,
                                                    const double4* __restrict__ vel, const double4* __restrict__ acc) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const double4 p = pos[k], v = vel[k], a = acc[k];
    double* o = aos + size_t(k) * stride_d;
    o[0] = p.x; o[1] = p.y; o[2] = p.z;
    o[3] = v.x; o[4] = v.y; o[5] = v.z;
    o[6] = a.x; o[7] = a.y; o[8] = a.z;
    o[9] = p.w;
}

// ---- K1: integrate_pre_force (shared.rs:135-140) + Bounds::contains (shared.rs:210-212)
__global__ __launch_bounds__(256) void k_drift_half(double4* __restrict__ pos, const double4* __restrict__ vel,
                                                    const int* __restrict__ count, unsigned char* __restrict__ keep,
                                                    int* __restrict__ escaped, double dt, Bounds64 b) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= *count) return;
    double4 p = pos[k];
    const double4 v = vel[k];
    p.x += (v.x * 0.5) * dt;
    p.y += (v.y * 0.5) * dt;
    p.z += (v.z * 0.5) * dt;
    pos[k] = p;
    const bool in = (p.x >= b.lo[0]) && (p.x <= b.hi[0]) && (p.y >= b.lo[1]) && (p.y <= b.hi[1]) && (p.z >= b.lo[2]) && (p.z <= b.hi[2]);
    keep[k] = in ? 1 : 0;
    if (!in) atomicAdd(escaped, 1);
}

// ---- K3: integrate_after_force (shared.rs:141-148)
__global__ __launch_bounds__(256) void k_kick_drift(double4* __restrict__ pos, double4* __restrict__ vel, const double4* __restrict__ acc,
                                                    const int* __restrict__ count, double dt) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= *count) return;
    double4 p = pos[k], v = vel[k];
    const double4 a = acc[k];
    v.x += a.x * dt;
    v.y += a.y * dt;
    v.z += a.z * dt;
    p.x += (v.x * 0.5) * dt;
    p.y += (v.y * 0.5) * dt;
    p.z += (v.z * 0.5) * dt;
    vel[k] = v;
    pos[k] = p;
}

// ---- K4: Vec::retain, one pass over many workgroups, in place (the scheme of kernels_integrate.hip k_compact)
constexpr int kTile = 1024;
constexpr unsigned long long kAgg = 1ull, kPrefix = 2ull;
__device__ __forceinline__ unsigned long long tile_word(int epoch, unsigned long long flag, int value) {
    return ((unsigned long long)(unsigned)epoch << 34) | (flag << 32) | (unsigned long long)(unsigned)value;
}

__global__ __launch_bounds__(kTile) void k_compact(double4* __restrict__ pos, double4* __restrict__ vel, double4* __restrict__ acc,
                                                   const unsigned char* __restrict__ keep, int* __restrict__ count,
                                                   int* __restrict__ escaped, unsigned long long* __restrict__ tile_state,
                                                   int* __restrict__ epoch_p) {
    if (*escaped == 0) return;
    __shared__ int wave_total[16];
    __shared__ int excl_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int n = *count;
    const int epoch = *epoch_p & 0x3fffffff;
    const int k = tile * kTile + tid;
    const bool kp = (k < n) && keep[k];
    double4 p = make_double4(0.0, 0.0, 0.0, 0.0), v = p, a = p;
    if (kp) { p = pos[k]; v = vel[k]; a = acc[k]; }
    const unsigned long long m = __ballot(kp);
    const int in_wave = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_total[wave] = __popcll(m);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's records are in registers before anything is published
    __syncthreads();
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
            st[0] = tile_word(epoch, kPrefix, total);
        } else {
            st[tile] = tile_word(epoch, kAgg, total);
            __threadfence();
            for (int j = tile - 1; j >= 0;) {
                const unsigned long long wd = st[j];
                if (int(wd >> 34) != epoch) continue;
                excl += int(unsigned(wd & 0xFFFFFFFFull));
                if (((wd >> 32) & 3ull) == kPrefix) break;
                --j;
            }
            st[tile] = tile_word(epoch, kPrefix, excl + total);
        }
        __threadfence();
        excl_s = excl;
        if (tile == int(gridDim.x) - 1) {
            *count = excl + total;
            *escaped = 0;
            *epoch_p = (epoch + 1) & 0x3fffffff;
        }
    }
    __syncthreads();
    if (kp) {
        const int d = excl_s + before + in_wave;
        pos[d] = p; vel[d] = v; acc[d] = a;
    }
}

// ---- K2: BruteForceSimulation::update_forces (brute_force.rs:64-82), one lane per body, partners in ascending order
constexpr int kStrictBlock = 256;
constexpr int kStrictTile = 512;   // 16 KB of LDS

__global__ __launch_bounds__(kStrictBlock) void k_bf_strict(const double4* __restrict__ pos_all, const int* __restrict__ seg_count, int n_seg,
                                                            int seg_cap, int my_seg, double4* __restrict__ acc, double g, double eps2,
                                                            unsigned long long* __restrict__ inter) {
    __shared__ double4 tile[kStrictTile];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kStrictBlock + tid;
    const int n = seg_count[my_seg];
    if (inter && blockIdx.x == 0 && tid == 0 && n > 0) {
        long long tot = 0;
        for (int sg = 0; sg < n_seg; ++sg) tot += seg_count[sg];
        atomicAdd(inter, (unsigned long long)n * (unsigned long long)(tot - 1));
    }
    const double4 pi = (i < n) ? pos_all[size_t(my_seg) * seg_cap + i] : make_double4(0.0, 0.0, 0.0, 0.0);
    double ax = 0.0, ay = 0.0, az = 0.0;  // :65-67
    // partners in ascending GLOBAL index: the blocks in rank order, each in its own order (a G-shard run adds in the
    // one-shard run's order: bit-equal)
    for (int sg = 0; sg < n_seg; ++sg) {
        const double4* __restrict__ pos = pos_all + size_t(sg) * seg_cap;
        const int m = seg_count[sg];
        for (int t0 = 0; t0 < m; t0 += kStrictTile) {
            const int cnt = min(kStrictTile, m - t0);
            __syncthreads();
            for (int k = tid; k < cnt; k += kStrictBlock) tile[k] = pos[t0 + k];
            __syncthreads();
            for (int j = 0; j < cnt; ++j) {
                if (sg == my_seg && t0 + j == i) continue;  // the reference never forms the i == j pair (:70-71)
                const double4 pj = tile[j];
                const double rx = pi.x - pj.x, ry = pi.y - pj.y, rz = pi.z - pj.z;   // :72
                const double r_dist = __builtin_sqrt((rx * rx + ry * ry) + rz * rz + eps2);  // :73
                const double r_cubed = r_dist * r_dist * r_dist;                      // :74
                const double force = (g / r_cubed);                                    // :77
                ax -= (rx * force) * pj.w;                                             // :78
                ay -= (ry * force) * pj.w;
                az -= (rz * force) * pj.w;
            }
        }
    }
    if (i < n) acc[i] = make_double4(ax, ay, az, 0.0);
}

// ---- K5: BarnesHutSimulation::calc_force (barnes_hut.rs:185-203) with the reference's nested sums (see
// kernels_bh.hip k_bh_walk_nested): the innermost open cell's sum in registers, the outer ones on a per-lane stack
constexpr int kWalkBlock = 64;
constexpr unsigned kCounterSlots = NBODY_WALK_COUNTER_SLOTS;

__device__ __forceinline__ void add_counts(unsigned long long* counters, unsigned n_acc, unsigned n_vis) {
    for (int off = 32; off > 0; off >>= 1) {
        n_acc += __shfl_down(n_acc, off);
        n_vis += __shfl_down(n_vis, off);
    }
    if ((threadIdx.x & 63) == 0 && counters) {
        const unsigned slot = blockIdx.x & (kCounterSlots - 1);
        atomicAdd(&counters[2 * slot], (unsigned long long)n_acc);
        atomicAdd(&counters[2 * slot + 1], (unsigned long long)n_vis);
    }
}

__global__ __launch_bounds__(kWalkBlock) void k_bh_walk_nested(const Node64* __restrict__ nodes, const int* __restrict__ order, int n_order,
                                                               const double4* __restrict__ pos, double4* __restrict__ acc, double g,
                                                               double eps2, double theta2, unsigned long long* __restrict__ counters,
                                                               Open64* __restrict__ stack, size_t stack_stride) {
    const int t = blockIdx.x * kWalkBlock + threadIdx.x;
    unsigned int n_acc = 0, n_vis = 0;
    if (t < n_order) {
        const int b = order[t];
        const double4 p = pos[b];
        double sx = 0.0, sy = 0.0, sz = 0.0;   // the innermost open cell's running sum
        int end = 0;                           // ... and the index after its subtree
        int d = 0;                             // open cells
        double ox = 0.0, oy = 0.0, oz = 0.0;   // calc_force(root)
        int i = 0;
        bool done = false;
        while (!done) {
            const Node64 nd = nodes[i];
            const double rx = nd.x - p.x, ry = nd.y - p.y, rz = nd.z - p.z;    // :190
            const double r2 = (rx * rx + ry * ry) + rz * rz;                     // :191
            const int skip = nd.skip;
            ++n_vis;
            double fx = 0.0, fy = 0.0, fz = 0.0;
            bool value = true;               // this visit yields a value for the enclosing cell
            if (nd.w2 < theta2 * r2) {                                           // :192
                const double r_dist = __builtin_sqrt(r2 + eps2);                 // :193
                const double r_cubed = r_dist * r_dist * r_dist;                 // :194
                const double k = ((g * nd.m) / r_cubed);                         // :195
                fx = rx * k; fy = ry * k; fz = rz * k;
                ++n_acc;
                i = skip;
            } else if (skip == i + 1) {      // a leaf (or the empty root): the fold of nothing is 0 (:197-202)
                i = skip;
            } else {                         // open the cell: its children fold into a fresh sum
                if (d > 0) stack[size_t(d) * stack_stride + t] = Open64{sx, sy, sz, end, 0};
                ++d;
                sx = sy = sz = 0.0;
                end = skip;
                i = i + 1;
                value = false;
            }
            if (value) {
                if (d == 0) { ox = fx; oy = fy; oz = fz; done = true; }        // the root itself was accepted (or is a leaf)
                else { sx += fx; sy += fy; sz += fz; }                          // acc += child result
            }
            while (!done && d > 0 && i == end) {   // the innermost cell is finished: hand its sum up
                const double vx = sx, vy = sy, vz = sz;
                --d;
                if (d == 0) { ox = vx; oy = vy; oz = vz; done = true; }
                else {
                    const Open64 up = stack[size_t(d) * stack_stride + t];
                    sx = up.x + vx; sy = up.y + vy; sz = up.z + vz;
                    end = up.end;
                }
            }
        }
        acc[b] = make_double4(ox, oy, oz, 0.0);  // :260
    }
    add_counts(counters, n_acc, n_vis);
}

// NBODY_LEAF_DIRECT: the walk of src/llm/barnes_hut.rs:915-997 on the same tree, one running sum in visit order
__global__ __launch_bounds__(kWalkBlock) void k_bh_walk_direct(const Node64* __restrict__ nodes, int n_nodes, const int* __restrict__ order,
                                                               int n_order, const double4* __restrict__ pos, double4* __restrict__ acc,
                                                               double g, double eps2, double theta2,
                                                               unsigned long long* __restrict__ counters) {
    const int t = blockIdx.x * kWalkBlock + threadIdx.x;
    unsigned int n_acc = 0, n_vis = 0;
    if (t < n_order) {
        const int b = order[t];
        const double4 p = pos[b];
        double ax = 0.0, ay = 0.0, az = 0.0;
        int i = 0;
        while (i < n_nodes) {
            const Node64 nd = nodes[i];
            const double rx = nd.x - p.x, ry = nd.y - p.y, rz = nd.z - p.z;
            const double r2 = (rx * rx + ry * ry) + rz * rz;
            ++n_vis;
            const int skip = nd.skip;
            if (r2 < 1e-10) { i = skip; continue; }                              // llm :933-935 (the body's own leaf: r2 = 0)
            if (nd.w2 < theta2 * r2 || skip == i + 1) {                          // llm :938 accepted cell, :958-972 leaf
                const double inv_r = 1.0 / __builtin_sqrt(r2 + eps2);            // llm :942
                const double inv_r3 = inv_r * inv_r * inv_r;                     // llm :944
                const double k = g * nd.m * inv_r3;                              // llm :947
                ax += rx * k; ay += ry * k; az += rz * k;                        // llm :950-952
                ++n_acc;
                i = skip;
            } else {
                i = i + 1;
            }
        }
        acc[b] = make_double4(ax, ay, az, 0.0);
    }
    add_counts(counters, n_acc, n_vis);
}

// ---- K5, fast arithmetic: the opening tests are the reference's (same products, same comparison: node counts stay
// exact on the host-built tree), the accepted monopoles are added into one running sum per lane with FMAs and 1/sqrt
__device__ __forceinline__ int walk_entry64(const Node64* __restrict__ nodes, const WalkSplit64& sp, int seg, const double4 p, double theta2, bool direct) {
    const int na = sp.n_anc[seg];
    for (int k = 0; k < na; ++k) {
        const Node64 nd = nodes[sp.anc[seg * kMaxAnc64 + k]];
        const double rx = nd.x - p.x, ry = nd.y - p.y, rz = nd.z - p.z;
        const double r2 = (rx * rx + ry * ry) + rz * rz;
        if (direct && r2 < 1e-10) return nd.skip;
        if (nd.w2 < theta2 * r2) return nd.skip;   // accepted: the walk resumes after its subtree
    }
    return sp.first[seg];
}

// BPL neighbouring bodies of the tree order per lane, walked in lockstep: the lane visits the smallest of their next
// indices, fetches that 64-byte record once and evaluates it for whichever bodies are due there (kernels_bh.hip
// k_bh_walk_duo: per body the same tests in the same order, the same sums; per lane the union of the sequences)
template <bool DIRECT, int BPL>
__global__ __launch_bounds__(kWalkBlock) void k_bh_walk_fast64(const Node64* __restrict__ nodes, int n_nodes, const int* __restrict__ order, int n_order,
                                                               const double4* __restrict__ pos, double4* __restrict__ acc, double g, double eps2,
                                                               double theta2, unsigned long long* __restrict__ counters, WalkSplit64 split, int xcd_blocks) {
    // (xcd_blocks != 0: XCD j -- workgroups j, j + 8, ... -- walks the j-th eighth of the tree order: kernels_bh.hip k_bh_walk_duo)
    const int bx = xcd_blocks ? int(blockIdx.x % 8) * xcd_blocks + int(blockIdx.x / 8) : int(blockIdx.x);
    const int t = bx * kWalkBlock + threadIdx.x;
    // a body group's long walks are in the segments around its own place in the tree: those first (kernels_bh.hip k_bh_walk)
    const int K = gridDim.y;
    const int diag = int((long long)bx * K / gridDim.x);
    const int kk = blockIdx.y;
    const int seg = ((diag + ((kk & 1) ? (kk + 1) / 2 : -(kk / 2))) % K + K) % K;
    const int s1 = split.first[seg + 1];
    unsigned int n_acc = 0, n_vis = 0;
    const int t0 = BPL * t;
    if (t0 < n_order) {
        double4 p[BPL];
        double ax[BPL], ay[BPL], az[BPL];
        int nx[BPL], body[BPL];
        int i = s1;
#pragma unroll
        for (int q = 0; q < BPL; ++q) {
            const bool live = t0 + q < n_order;
            body[q] = order[live ? t0 + q : t0];
            p[q] = pos[body[q]];
            ax[q] = ay[q] = az[q] = 0.0;
            nx[q] = live ? walk_entry64(nodes, split, seg, p[q], theta2, DIRECT) : s1;
            i = min(i, nx[q]);
        }
        while (i < s1) {
            const Node64 nd = nodes[i];
            const int skip = nd.skip;
            int nxt = s1;
#pragma unroll
            for (int q = 0; q < BPL; ++q) {
                if (nx[q] == i) {
                    const double rx = nd.x - p[q].x, ry = nd.y - p[q].y, rz = nd.z - p[q].z;
                    const double r2 = (rx * rx + ry * ry) + rz * rz;
                    ++n_vis;
                    if (DIRECT && r2 < 1e-10) nx[q] = skip;
                    else if (nd.w2 < theta2 * r2 || (DIRECT && skip == i + 1)) {
                        const double rinv = rsqrt(r2 + eps2);
                        const double k = (g * nd.m) * ((rinv * rinv) * rinv);
                        ax[q] = fma(rx, k, ax[q]); ay[q] = fma(ry, k, ay[q]); az[q] = fma(rz, k, az[q]);
                        ++n_acc;
                        nx[q] = skip;
                    } else {
                        nx[q] = i + 1;
                    }
                }
                nxt = min(nxt, nx[q]);
            }
            i = nxt;
        }
#pragma unroll
        for (int q = 0; q < BPL; ++q)
            if (t0 + q < n_order)
                *(split.n_seg > 1 ? split.planes + size_t(seg) * split.plane_stride + (t0 + q) : acc + body[q]) = make_double4(ax[q], ay[q], az[q], 0.0);
    }
    add_counts(counters, n_acc, n_vis);
}

// KICK: integrate_after_force (shared.rs:141-148) rides along (k_kick_drift's arithmetic, one launch less per step)
template <bool KICK>
__global__ __launch_bounds__(256) void k_bh_reduce64(const double4* __restrict__ planes, int n_seg, size_t plane_stride, const int* __restrict__ order,
                                                     int n_order, double4* __restrict__ acc, double4* __restrict__ pos, double4* __restrict__ vel, double dt) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_order) return;
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int k = 0; k < n_seg; ++k) {   // segment order = the order the single walk adds them in
        const double4 v = planes[size_t(k) * plane_stride + t];
        sx += v.x; sy += v.y; sz += v.z;
    }
    const int b = order[t];
    acc[b] = make_double4(sx, sy, sz, 0.0);
    if (KICK) {
        double4 p = pos[b], v = vel[b];
        v.x += sx * dt;
        v.y += sy * dt;
        v.z += sz * dt;
        p.x += (v.x * 0.5) * dt;
        p.y += (v.y * 0.5) * dt;
        p.z += (v.z * 0.5) * dt;
        vel[b] = v;
        pos[b] = p;
    }
}

// ---- diagnostics: KE and pair-potential row sums, per-block partials {KE, sum_j m_i m_j / d_ij}
constexpr int kEnergyBlock = 256;
__global__ __launch_bounds__(kEnergyBlock) void k_energy(const double4* __restrict__ pos, const double4* __restrict__ vel,
                                                         const int* __restrict__ count, double eps2, double* __restrict__ out2) {
    __shared__ double4 tile[kEnergyBlock];
    __shared__ double red[2][kEnergyBlock];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kEnergyBlock + tid;
    const int n = *count;
    const double4 pi = (i < n) ? pos[i] : make_double4(0.0, 0.0, 0.0, 0.0);
    double ke = 0.0, pe = 0.0;
    if (i < n) { const double4 v = vel[i]; ke = 0.5 * pi.w * ((v.x * v.x + v.y * v.y) + v.z * v.z); }
    for (int t0 = 0; t0 < n; t0 += kEnergyBlock) {
        __syncthreads();
        tile[tid] = (t0 + tid < n) ? pos[t0 + tid] : make_double4(0.0, 0.0, 0.0, 0.0);
        __syncthreads();
        const int cnt = min(kEnergyBlock, n - t0);
        if (i < n)
            for (int j = 0; j < cnt; ++j) {
                if (t0 + j == i) continue;
                const double4 pj = tile[j];
                const double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                pe += pi.w * pj.w / sqrt(dx * dx + dy * dy + dz * dz + eps2);
            }
    }
    red[0][tid] = ke; red[1][tid] = pe;
    __syncthreads();
    for (int off = kEnergyBlock / 2; off > 0; off >>= 1) {
        if (tid < off) { red[0][tid] += red[0][tid + off]; red[1][tid] += red[1][tid + off]; }
        __syncthreads();
    }
    if (tid == 0) { out2[2 * blockIdx.x] = red[0][0]; out2[2 * blockIdx.x + 1] = red[1][0]; }
}

inline int blocks_for(int n, int bs) { return n <= 0 ? 0 : (n + bs - 1) / bs; }

}  // namespace

void launch_aos_to_soa(hipStream_t s, const double* aos, int stride_d, int n, const Dev& d, size_t first) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_aos_to_soa, dim3(blocks_for(n, 256)), dim3(256), 0, s, aos, stride_d, n, d.pos + first, d.vel + first, d.acc + first);
}
void launch_aos_to_pos(hipStream_t s, const double* aos, int stride_d, int n, double4* pos) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_aos_to_pos, dim3(blocks_for(n, 256)), dim3(256), 0, s, aos, stride_d, n, pos);
}
void launch_soa_to_aos(hipStream_t s, double* aos, int stride_d, int n, const Dev& d) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_soa_to_aos, dim3(blocks_for(n, 256)), dim3(256), 0, s, aos, stride_d, n, d.pos, d.vel, d.acc);
}
void launch_drift_half(hipStream_t s, const Dev& d, int n_upper, double dt, const Bounds64& b) {
    if (n_upper <= 0) return;
    hipLaunchKernelGGL(k_drift_half, dim3(blocks_for(n_upper, 256)), dim3(256), 0, s, d.pos, d.vel, d.count, d.keep, d.escaped, dt, b);
}
void launch_compact(hipStream_t s, const Dev& d, int n_upper) {
    if (n_upper <= 0) return;
    hipLaunchKernelGGL(k_compact, dim3(blocks_for(n_upper, kTile)), dim3(kTile), 0, s, d.pos, d.vel, d.acc, d.keep, d.count, d.escaped,
                       d.tile_state, d.epoch);
}
void launch_kick_drift(hipStream_t s, const Dev& d, int n_upper, double dt) {
    if (n_upper <= 0) return;
    hipLaunchKernelGGL(k_kick_drift, dim3(blocks_for(n_upper, 256)), dim3(256), 0, s, d.pos, d.vel, d.acc, d.count, dt);
}
void launch_bf_strict(hipStream_t s, const Dev& d, int n_upper, double g, double eps2) {
    if (n_upper <= 0) return;
    hipLaunchKernelGGL(k_bf_strict, dim3(blocks_for(n_upper, kStrictBlock)), dim3(kStrictBlock), 0, s, d.pos_all, d.seg_count, d.n_seg, d.cap, d.my_seg,
                       d.acc, g, eps2, d.inter);
}
void launch_bh_walk(hipStream_t s, const Dev& d, const Node64* nodes, int n_nodes, const int* order, int n_order, double g, double eps2,
                    double theta2, unsigned long long* counters, int leaf_direct, Open64* stack, size_t stack_stride) {
    if (n_order <= 0) return;
    const dim3 grid(blocks_for(n_order, kWalkBlock));
    if (leaf_direct)
        hipLaunchKernelGGL(k_bh_walk_direct, grid, dim3(kWalkBlock), 0, s, nodes, n_nodes, order, n_order, d.pos, d.acc, g, eps2, theta2, counters);
    else
        hipLaunchKernelGGL(k_bh_walk_nested, grid, dim3(kWalkBlock), 0, s, nodes, order, n_order, d.pos, d.acc, g, eps2, theta2, counters,
                           stack, stack_stride);
}
void launch_bh_walk_fast(hipStream_t s, const Dev& d, const Node64* nodes, int n_nodes, const int* order, int n_order, double g, double eps2,
                         double theta2, unsigned long long* counters, int leaf_direct, const WalkSplit64& split, int bodies_per_lane, const double* kick_dt, int* kicked) {
    if (kicked) *kicked = 0;
    if (n_order <= 0) return;
    const int bpl = bodies_per_lane >= 6 ? 6 : bodies_per_lane >= 4 ? 4 : bodies_per_lane == 3 ? 3 : bodies_per_lane == 2 ? 2 : 1;
    const int gx = int(blocks_for((n_order + bpl - 1) / bpl, kWalkBlock)), gx8 = (gx + 7) / 8 * 8;
    const int xcd_blocks = nbody::tuning().bh_walk_xcd ? gx8 / 8 : 0;
    const dim3 grid(xcd_blocks ? gx8 : gx, split.n_seg);
#define WALK64(D, B) hipLaunchKernelGGL((k_bh_walk_fast64<D, B>), grid, dim3(kWalkBlock), 0, s, nodes, n_nodes, order, n_order, d.pos, d.acc, g, eps2, theta2, counters, split, xcd_blocks)
#define WALK64_B(D) do { if (bpl == 6) WALK64(D, 6); else if (bpl == 4) WALK64(D, 4); else if (bpl == 3) WALK64(D, 3); else if (bpl == 2) WALK64(D, 2); else WALK64(D, 1); } while (0)
    if (leaf_direct) WALK64_B(true); else WALK64_B(false);
#undef WALK64_B
#undef WALK64
    if (split.n_seg > 1) {
        if (kick_dt) {   // (every own body is in `order` exactly once: the kick reaches them all)
            hipLaunchKernelGGL(k_bh_reduce64<true>, dim3(blocks_for(n_order, 256)), dim3(256), 0, s, split.planes, split.n_seg, split.plane_stride, order, n_order,
                               d.acc, d.pos, d.vel, *kick_dt);
            if (kicked) *kicked = 1;
        } else {
            hipLaunchKernelGGL(k_bh_reduce64<false>, dim3(blocks_for(n_order, 256)), dim3(256), 0, s, split.planes, split.n_seg, split.plane_stride, order, n_order,
                               d.acc, d.pos, d.vel, 0.0);
        }
    }
}
void launch_energy(hipStream_t s, const Dev& d, int n_upper, double eps2, double* out2) {
    if (n_upper <= 0) return;
    hipLaunchKernelGGL(k_energy, dim3(blocks_for(n_upper, kEnergyBlock)), dim3(kEnergyBlock), 0, s, d.pos, d.vel, d.count, eps2, out2);
}

}  // namespace nbody64
