// kernels_bf.hip -- K2: all-pairs gravity, BruteForceSimulation::update_forces
// (src/manual/brute_force.rs:64-82), for gfx950.
//
// The reference walks the unordered pairs (i > j) once and applies Newton's third law; body k
// therefore meets its partners in ascending index order, a_k -= ((p_k - p_j) * f) * m_j for j < k
// and a_k += ((p_i - p_k) * f) * m_i for i > k.  Negation is exact in IEEE arithmetic, so both
// branches are the single expression a_k -= ((p_k - p_p) * f) * m_p over p != k, p ascending.
//
//   strict : one lane per body, partners streamed in ascending order through an LDS tile,
//            d = sqrt((x*x + y*y) + z*z + eps^2), f = g / ((d*d)*d), no contraction, correctly
//            rounded sqrt/divide  -> the reference's rounding sequence, bit for bit.
//   fast   : IPT bodies per lane in registers, the partner tile split over the workgroup's waves,
//            v_rsq_f32 + FMA, partial accelerations of the waves combined through LDS in a fixed
//            order (deterministic, but a different summation order: <= 1e-5 relative).
//
// Both are fp32-VALU bound (20 flop per directed pair against 16 B of HBM traffic per body):
// partner positions are read once per workgroup from L2 and broadcast from LDS with
// ds_read_b128, one LDS instruction per 64*IPT pair evaluations.
#include "kernels.h"

namespace nbody {

// NbodyStats::interactions of a brute-force pass: n_own * (n_total - 1) from the live counts (one thread per launch)
__device__ __forceinline__ void count_interactions(unsigned long long* inter, const int* seg_count, int n_seg, int n_own) {
    long long tot = 0;
    for (int s = 0; s < n_seg; ++s) tot += seg_count[s];
    if (tot > 0) atomicAdd(inter, (unsigned long long)n_own * (unsigned long long)(tot - 1));
}

// ------------------------------------------------------------------------------------ strict
constexpr int kStrictBlock = 256;
constexpr int kStrictTile = 1024;

__global__ __launch_bounds__(kStrictBlock) void k_bf_strict(const float4* __restrict__ pos_all,
                                                            const int* __restrict__ seg_count, int n_seg, int seg_cap,
                                                            int my_seg, float4* __restrict__ acc, float g, float eps2,
                                                            unsigned long long* __restrict__ inter) {
    __shared__ float4 tile[kStrictTile];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kStrictBlock + tid;
    const int n_own = seg_count[my_seg];
    if (inter && blockIdx.x == 0 && tid == 0) count_interactions(inter, seg_count, n_seg, n_own);
    const float4 pi = (i < n_own) ? pos_all[size_t(my_seg) * seg_cap + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float ax = 0.f, ay = 0.f, az = 0.f;  // brute_force.rs:65-67
    for (int s = 0; s < n_seg; ++s) {
        const int ns = seg_count[s];
        const float4* __restrict__ ps = pos_all + size_t(s) * seg_cap;
        const int self = (s == my_seg) ? i : -1;
        for (int t0 = 0; t0 < ns; t0 += kStrictTile) {
            const int cnt = min(kStrictTile, ns - t0);
            __syncthreads();
            for (int k = tid; k < cnt; k += kStrictBlock) tile[k] = ps[t0 + k];
            __syncthreads();
            for (int j = 0; j < cnt; ++j) {
                if (t0 + j == self) continue;  // the reference never forms the i == j pair (:70-71)
                const float4 pj = tile[j];
                const float rx = pi.x - pj.x, ry = pi.y - pj.y, rz = pi.z - pj.z;  // :72
                const float r_dist = __builtin_sqrtf((rx * rx + ry * ry) + rz * rz + eps2);  // :73
                const float r_cubed = r_dist * r_dist * r_dist;                    // :74
                const float force = (g / r_cubed);                         // :77
                ax -= (rx * force) * pj.w;                                         // :78
                ay -= (ry * force) * pj.w;
                az -= (rz * force) * pj.w;
            }
        }
    }
    if (i < n_own) acc[i] = make_float4(ax, ay, az, 0.f);
}

// -------------------------------------------------------------------------------------- fast
// Workgroup = WAVES waves that all hold the same 64*IPT bodies (lane l: bodies base + q*64 + l)
// and each take 1/WAVES of every partner tile.
template <int IPT, int WAVES, int TILE>
__global__ __launch_bounds__(WAVES * 64) void k_bf_fast(const float4* __restrict__ pos_all,
                                                        const int* __restrict__ seg_count, int n_seg, int seg_cap,
                                                        int my_seg, float4* __restrict__ acc, float g, float eps2,
                                                        unsigned long long* __restrict__ inter) {
    constexpr int NT = WAVES * 64;
    constexpr int SLICE = TILE / WAVES;
    static_assert(TILE % WAVES == 0, "tile must split evenly over the waves");
    static_assert(TILE * 4 >= WAVES * IPT * 3 * 64, "reduction scratch must fit in the tile");
    __shared__ float4 tile[TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ibase = blockIdx.x * (64 * IPT);
    const int n_own = seg_count[my_seg];
    const float4* __restrict__ own = pos_all + size_t(my_seg) * seg_cap;
    if (inter && blockIdx.x == 0 && tid == 0) count_interactions(inter, seg_count, n_seg, n_own);

    float px[IPT], py[IPT], pz[IPT], ax[IPT], ay[IPT], az[IPT];
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
        const int i = ibase + q * 64 + lane;
        const float4 p = (i < n_own) ? own[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        px[q] = p.x; py[q] = p.y; pz[q] = p.z;
        ax[q] = ay[q] = az[q] = 0.f;
    }

    for (int s = 0; s < n_seg; ++s) {
        const int ns = seg_count[s];
        const float4* __restrict__ ps = pos_all + size_t(s) * seg_cap;
        for (int t0 = 0; t0 < ns; t0 += TILE) {
            const int cnt = min(TILE, ns - t0);
            __syncthreads();
            for (int k = tid; k < cnt; k += NT) tile[k] = ps[t0 + k];
            __syncthreads();
            const int j0 = wave * SLICE;
            const int j1 = min(cnt, j0 + SLICE);
            // does this wave's partner slice contain any of the workgroup's own bodies?
            const bool diag = (s == my_seg) && (t0 + j0 < ibase + 64 * IPT) && (t0 + j1 > ibase);
            if (!diag) {
#pragma unroll 4
                for (int j = j0; j < j1; ++j) {
                    const float4 pj = tile[j];
#pragma unroll
                    for (int q = 0; q < IPT; ++q) {
                        const float dx = pj.x - px[q], dy = pj.y - py[q], dz = pj.z - pz[q];
                        const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, eps2)));
                        const float rinv = __builtin_amdgcn_rsqf(r2);
                        const float sc = (pj.w * rinv) * (rinv * rinv);
                        ax[q] = __builtin_fmaf(dx, sc, ax[q]);
                        ay[q] = __builtin_fmaf(dy, sc, ay[q]);
                        az[q] = __builtin_fmaf(dz, sc, az[q]);
                    }
                }
            } else {
                for (int j = j0; j < j1; ++j) {
                    const float4 pj = tile[j];
#pragma unroll
                    for (int q = 0; q < IPT; ++q) {
                        const float dx = pj.x - px[q], dy = pj.y - py[q], dz = pj.z - pz[q];
                        const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, eps2)));
                        const float rinv = __builtin_amdgcn_rsqf(r2);
                        float sc = (pj.w * rinv) * (rinv * rinv);
                        sc = (t0 + j == ibase + q * 64 + lane) ? 0.f : sc;  // no self pair (eps may be 0)
                        ax[q] = __builtin_fmaf(dx, sc, ax[q]);
                        ay[q] = __builtin_fmaf(dy, sc, ay[q]);
                        az[q] = __builtin_fmaf(dz, sc, az[q]);
                    }
                }
            }
        }
    }

    // combine the waves' partial sums in wave order through LDS
    __syncthreads();
    float* red = reinterpret_cast<float*>(tile);  // [WAVES][IPT*3][64]
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
        red[((wave * IPT + q) * 3 + 0) * 64 + lane] = ax[q];
        red[((wave * IPT + q) * 3 + 1) * 64 + lane] = ay[q];
        red[((wave * IPT + q) * 3 + 2) * 64 + lane] = az[q];
    }
    __syncthreads();
    for (int q = wave; q < IPT; q += WAVES) {
        float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            sx += red[((w * IPT + q) * 3 + 0) * 64 + lane];
            sy += red[((w * IPT + q) * 3 + 1) * 64 + lane];
            sz += red[((w * IPT + q) * 3 + 2) * 64 + lane];
        }
        const int i = ibase + q * 64 + lane;
        if (i < n_own) acc[i] = make_float4(g * sx, g * sy, g * sz, 0.f);
    }
}

void launch_bf_forces_strict(hipStream_t s, const Shard& sh, int n_upper, float g, float g_soft2) {
    if (n_upper <= 0) return;
    int blocks = (n_upper + kStrictBlock - 1) / kStrictBlock;
    hipLaunchKernelGGL(k_bf_strict, dim3(blocks), dim3(kStrictBlock), 0, s, sh.pos_all, sh.seg_count, sh.n_seg,
                       sh.seg_cap, sh.my_seg, sh.acc, g, g_soft2, sh.inter);
}

template <int IPT, int WAVES, int TILE>
static void launch_fast_cfg(hipStream_t s, const Shard& sh, int n_upper, float g, float eps2) {
    int blocks = (n_upper + 64 * IPT - 1) / (64 * IPT);
    hipLaunchKernelGGL((k_bf_fast<IPT, WAVES, TILE>), dim3(blocks), dim3(WAVES * 64), 0, s, sh.pos_all,
                       sh.seg_count, sh.n_seg, sh.seg_cap, sh.my_seg, sh.acc, g, eps2, sh.inter);
}

}  // namespace nbody
// 0 = default; != 0 forces the LDS-tiled one-sided kernel with that many bodies per lane (1, 2, 4) even
// where the symmetric kernel would be used: tuning/test hook (NBODY_BF_VARIANT environment variable)
namespace nbody {

void launch_bf_forces_fast(hipStream_t s, const Shard& sh, int n_upper, float g, float g_soft2) {
    if (n_upper <= 0) return;
    // one body per lane, 8 waves splitting each partner tile, measured fastest at every size once
    // SLP vectorisation was off (tools/tune_bf.py); 2 and 4 bodies per lane stay selectable
    switch (tuning().bf_fast_variant) {
        case 4: launch_fast_cfg<4, 8, 2048>(s, sh, n_upper, g, g_soft2); break;
        case 2: launch_fast_cfg<2, 8, 2048>(s, sh, n_upper, g, g_soft2); break;
        default: launch_fast_cfg<1, 8, 2048>(s, sh, n_upper, g, g_soft2); break;
    }
}

// ---------------------------------------------------------------------------- f64 diagnostics
// KE and pair-potential row sums of the own segment; per-block partials summed on the host in
// block order (deterministic).  Not a reference function.
constexpr int kEnergyBlock = 256;
__global__ __launch_bounds__(kEnergyBlock) void k_energy(const float4* __restrict__ pos_all,
                                                         const float4* __restrict__ vel,
                                                         const int* __restrict__ seg_count, int n_seg, int seg_cap,
                                                         int my_seg, double eps2, double* __restrict__ partial) {
    __shared__ float4 tile[kEnergyBlock];
    __shared__ double red[2][kEnergyBlock / 64];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kEnergyBlock + tid;
    const int n_own = seg_count[my_seg];
    const bool live = i < n_own;
    const float4 pi = live ? pos_all[size_t(my_seg) * seg_cap + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    double u = 0.0;
    for (int s = 0; s < n_seg; ++s) {
        const int ns = seg_count[s];
        const float4* __restrict__ ps = pos_all + size_t(s) * seg_cap;
        const int self = (s == my_seg) ? i : -1;
        for (int t0 = 0; t0 < ns; t0 += kEnergyBlock) {
            const int cnt = min(kEnergyBlock, ns - t0);
            __syncthreads();
            if (tid < cnt) tile[tid] = ps[t0 + tid];
            __syncthreads();
            for (int j = 0; j < cnt; ++j) {
                if (t0 + j == self) continue;
                const float4 pj = tile[j];
                const double dx = double(pi.x) - double(pj.x), dy = double(pi.y) - double(pj.y),
                             dz = double(pi.z) - double(pj.z);
                u += double(pj.w) / sqrt(dx * dx + dy * dy + dz * dz + eps2);
            }
        }
    }
    double ke = 0.0, pe = 0.0;
    if (live) {
        const float4 v = vel[i];
        ke = 0.5 * double(pi.w) * (double(v.x) * v.x + double(v.y) * v.y + double(v.z) * v.z);
        pe = u * double(pi.w);
    }
    for (int off = 32; off > 0; off >>= 1) {
        ke += __shfl_down(ke, off);
        pe += __shfl_down(pe, off);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = ke; red[1][tid >> 6] = pe; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < kEnergyBlock / 64; ++w) { a += red[0][w]; b += red[1][w]; }
        partial[2 * blockIdx.x + 0] = a;
        partial[2 * blockIdx.x + 1] = b;
    }
}

void launch_energy(hipStream_t s, const Shard& sh, int n_upper, double g_soft2, double* partial) {
    if (n_upper <= 0) return;
    int blocks = (n_upper + kEnergyBlock - 1) / kEnergyBlock;
    hipLaunchKernelGGL(k_energy, dim3(blocks), dim3(kEnergyBlock), 0, s, sh.pos_all, sh.vel, sh.seg_count, sh.n_seg,
                       sh.seg_cap, sh.my_seg, g_soft2, partial);
}

}  // namespace nbody
