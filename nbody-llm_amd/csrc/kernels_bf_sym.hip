// kernels_bf_sym.hip -- K2 "symmetric": all-pairs gravity that evaluates every unordered pair
// ONCE and applies it to both bodies, as BruteForceSimulation::update_forces does on the CPU
// (src/manual/brute_force.rs:70-81: a_i -= r f m_j ; a_j += r f m_i), restated for 64-lane waves.
//
// Work unit = one wave; no LDS memory, no barrier:
//   * the wave keeps a RESIDENT SET of 64*IPT bodies in registers (lane l: bodies q*64 + l);
//   * TRAVELLING CHUNKS of 64 bodies (one per lane, with their own accumulators) are rotated
//     through the lanes one lane per step with ds_bpermute_b32 (the LDS crossbar, which runs beside
//     the VALU; a DPP wave_ror:1 rotate costs ~10 VALU cycles per register in this stream); after
//     64 steps every lane has met every travelling body and the chunk is back in its home lanes;
//   * each step evaluates IPT pairs per lane and updates both sides: 16 VALU + 1 v_rsq_f32 per
//     unordered pair (= 2 directed interactions), written stage by stage over the IPT pairs so
//     that no instruction waits for its predecessor.
// Measured on MI355X (tools/sym_cycles.py, in-kernel stamps), packed form (the default): 313 SIMD
// cycles per 8-pair step at the 2.06 GHz the chip holds under v_pk_* = 64 packed x ~4.3 + 8 x ~8
// (v_rsq_f32); the scalar form (NBODY_SYM_PACKED=0) takes 388 at 2.37 GHz.  fp32-issue bound either way.
//
// Decomposition (balanced by construction): with A resident sets, set a meets the chunks of sets
// a+1 .. a+ceil(A/2)-1 (cyclic) symmetrically (k_bf_sym); the pairs inside a set and, for even
// A, with the opposite set are evaluated one-sidedly by k_bf_sym_rest (~3 % of the evaluations).
// The chunk sequence of a set is cut into K equal parts, one per wave.
// Partial sums go to "planes" of float4[n_pad]: plane d-1 receives the travelling-side sums of
// set distance d, plane sym_sets+p the resident-side sums of slice p of each set, the last
// plane k_bf_sym_rest's; every plane entry is written exactly once per launch, so the reduction
// (k_bf_sym_reduce) adds the planes in a fixed order: deterministic, no atomics.
//
// fast math only (v_rsq_f32 + FMA; summation order differs from the reference): tolerance 1e-5.
#include "kernels.h"
#include "bf_pair.h"

#include <algorithm>

namespace nbody {

namespace {

#ifdef NBODY_TUNING
__device__ unsigned long long nbody_sym_stamps[3 * 8192];  // diagnostic builds only (DBG & 4)
#endif

}  // namespace

template <int IPT>
__device__ __forceinline__ void sym_rest_wave(const float4* __restrict__ pos, int n, int A, int group, int slice,
                                              int n_slices, float& ax, float& ay, float& az, float eps2);

template <int IPT, int NW>
__device__ __forceinline__ void sym_rest_block(const float4* __restrict__ pos, int n, int A, int group,
                                               float4* __restrict__ plane, float eps2, float (*red)[3][64]);

// WPB waves per workgroup, sized so that exactly one (WPB = 16) or two (WPB = 12) workgroups fit a
// CU: the dispatcher then has no choice but to spread the grid evenly.  With small workgroups it
// packs some CUs to their register limit and leaves others nearly empty, and the kernel lasts as
// long as the fullest SIMD (measured with in-kernel stamps: wave lifetimes 560-960 us, +55 %).
template <int IPT, int WPB, int DBG = 0, bool PK = false>
__global__ __launch_bounds__(WPB * 64) void k_bf_sym(const float4* __restrict__ pos, const int* __restrict__ count,
                                                     int A, int K, const int* __restrict__ bounds, int sym_sets,
                                                     float4* __restrict__ planes, size_t plane_stride, float eps2,
                                                     int res_combine) {
    const int lane = threadIdx.x & 63;
    // Slices of a set differ by one chunk (e.g. 15,16,16,16,15,...).  A workgroup's waves land on the
    // SIMDs cyclically (wave w on SIMD w % 4, observed via HW_ID), so slices are dealt such that each
    // SIMD gets WPB/4 CONSECUTIVE slices -- one short one and the long ones -- and all four SIMDs of
    // the CU carry the same number of chunks.
    const int wslot = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_main_blocks = (A * K + WPB - 1) / WPB;
    if (int(blockIdx.x) >= n_main_blocks) {
        // tail filler: the workgroups after the rotation workgroups are dispatched as CUs fall free
        // and do the own-set / opposite-set pairs (64 bodies per workgroup) inside the same launch
        __shared__ float red[WPB - 1][3][64];
        sym_rest_block<IPT, WPB>(pos, *count, A, blockIdx.x - n_main_blocks,
                                 planes + size_t(sym_sets + (res_combine ? K / WPB : K)) * plane_stride, eps2, red);
        return;
    }
    const int gw = blockIdx.x * WPB + (wslot & 3) * (WPB / 4) + (wslot >> 2);  // global slice index
    if (gw >= A * K) return;
    const int a = gw / K;                        // resident set
    const int part = gw - a * K;                 // this wave's slice of the set's chunk sequence
    const int n = *count;
    const int Cn = A * IPT;                      // chunks in the padded body array
    const int k0 = bounds[part], k1 = bounds[part + 1];
    const int src_lane = ((lane + 63) & 63) * 4;  // ds_bpermute address: take from the lane below
    const int src_lane2 = ((lane + 62) & 63) * 4; // two lanes below
    // eps2 lives in a VGPR: with an SGPR operand the compiler cannot tell inside the loop whether the
    // kernel-argument load has landed and drains ALL pending LDS traffic (lgkmcnt(0)) every step
    float eps2v = eps2;
    asm volatile("" : "+v"(eps2v));

    float xi[IPT], yi[IPT], zi[IPT], mi[IPT], axi[IPT], ayi[IPT], azi[IPT];
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
        const int i = (a * IPT + q) * 64 + lane;
        const float4 p = (i < n) ? pos[i] : make_float4(PAD_POS, PAD_POS, PAD_POS, 0.f);
        xi[q] = p.x; yi[q] = p.y; zi[q] = p.z; mi[q] = p.w;
        axi[q] = ayi[q] = azi[q] = 0.f;
    }

    // chunk k of this set's sequence: the chunks of the next sym_sets sets, in cyclic order
    auto chunk_of = [&](int k) {
        int c = (a + 1) * IPT + k;
        if (c >= Cn) c -= Cn;
        return c;
    };
    auto load_chunk = [&](int k) {
        const int j = chunk_of(k) * 64 + lane;
        return (j < n) ? pos[j] : make_float4(PAD_POS, PAD_POS, PAD_POS, 0.f);
    };

    [[maybe_unused]] unsigned long long t_beg = 0, r_beg = 0;
    if (DBG & 4) { t_beg = __builtin_amdgcn_s_memtime(); r_beg = __builtin_amdgcn_s_memrealtime(); }
    float4 nxt = (k0 < k1) ? load_chunk(k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    // packed form (PK): the resident bodies two per register pair, see bf_pair.h
    v2f xi2[IPT / 2], yi2[IPT / 2], zi2[IPT / 2], mi2[IPT / 2], axi2[IPT / 2], ayi2[IPT / 2], azi2[IPT / 2];
    if (PK) {
        pack_pairs<IPT>(xi, xi2); pack_pairs<IPT>(yi, yi2); pack_pairs<IPT>(zi, zi2); pack_pairs<IPT>(mi, mi2);
        pack_pairs<IPT>(axi, axi2); pack_pairs<IPT>(ayi, ayi2); pack_pairs<IPT>(azi, azi2);
    }
    for (int k = k0; k < k1; ++k) {
        float xj = nxt.x, yj = nxt.y, zj = nxt.z, mj = nxt.w;
        float axj = 0.f, ayj = 0.f, azj = 0.f;
        v2f axj2 = {0.f, 0.f}, ayj2 = {0.f, 0.f}, azj2 = {0.f, 0.f};
        if (k + 1 < k1) nxt = load_chunk(k + 1);
        // The SIMD arbitrates VALU issue by priority, then age: at equal priority the oldest wave
        // runs at full speed, the youngest gets the leftovers, and once the old waves are done the
        // last one runs alone at half the SIMD's rate.  Priority = quarter of the work still to
        // do, so the waves of a SIMD advance together and finish together.
        switch ((4 * (k1 - k) - 1) / (k1 - k0)) {
            case 3: __builtin_amdgcn_s_setprio(3); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(0); break;
        }
        if (PK) {
            // the next step's positions are requested at the start of this one (they arrive long before
            // its end); two steps per iteration so that the two register sets swap roles without moves
            for (int s = 0; s < 64; s += 2) {
                const float x1 = rotl(xj, src_lane), y1 = rotl(yj, src_lane), z1 = rotl(zj, src_lane), m1 = rotl(mj, src_lane);
                pair_evals_pk<IPT / 2, DBG>(xi2, yi2, zi2, mi2, axi2, ayi2, azi2, xj, yj, zj, mj, axj2, ayj2, azj2, eps2v);
                // the accumulators follow their body; they are next needed at the END of the next step
                axj2.x = rotl(axj2.x, src_lane); ayj2.x = rotl(ayj2.x, src_lane); azj2.x = rotl(azj2.x, src_lane);
                axj2.y = rotl(axj2.y, src_lane); ayj2.y = rotl(ayj2.y, src_lane); azj2.y = rotl(azj2.y, src_lane);
                xj = rotl(x1, src_lane); yj = rotl(y1, src_lane); zj = rotl(z1, src_lane); mj = rotl(m1, src_lane);
                pair_evals_pk<IPT / 2, DBG>(xi2, yi2, zi2, mi2, axi2, ayi2, azi2, x1, y1, z1, m1, axj2, ayj2, azj2, eps2v);
                axj2.x = rotl(axj2.x, src_lane); ayj2.x = rotl(ayj2.x, src_lane); azj2.x = rotl(azj2.x, src_lane);
                axj2.y = rotl(axj2.y, src_lane); ayj2.y = rotl(ayj2.y, src_lane); azj2.y = rotl(azj2.y, src_lane);
            }
        } else {
            // positions are rotated two steps ahead (they do not change while the chunk travels), so the
            // crossbar latency never sits between a rotate and its first use
            float x1 = rotl(xj, src_lane), y1 = rotl(yj, src_lane), z1 = rotl(zj, src_lane), m1 = rotl(mj, src_lane);
#pragma unroll 2
            for (int s = 0; s < 64; ++s) {
                const float x2 = rotl(xj, src_lane2), y2 = rotl(yj, src_lane2), z2 = rotl(zj, src_lane2), m2 = rotl(mj, src_lane2);
                pair_evals<IPT, DBG>(xi, yi, zi, mi, axi, ayi, azi, xj, yj, zj, mj, axj, ayj, azj, eps2v);
                axj = rotl(axj, src_lane); ayj = rotl(ayj, src_lane); azj = rotl(azj, src_lane);
                xj = x1; yj = y1; zj = z1; mj = m1;
                x1 = x2; y1 = y2; z1 = z2; m1 = m2;
            }
        }
        if (PK) { axj = axj2.x + axj2.y; ayj = ayj2.x + ayj2.y; azj = azj2.x + azj2.y; }
        const int d = k / IPT + 1;  // set distance 1..sym_sets
        planes[size_t(d - 1) * plane_stride + size_t(chunk_of(k)) * 64 + lane] = make_float4(axj, ayj, azj, 0.f);
    }
    if (PK) { unpack_pairs<IPT>(axi2, axi); unpack_pairs<IPT>(ayi2, ayi); unpack_pairs<IPT>(azi2, azi); }

#ifdef NBODY_TUNING
    if (DBG & 4) {  // diagnostic build: shader cycles and 100 MHz ticks of this wave's chunk loop, into a
                    // buffer of their own that nothing else reads
        const unsigned long long t_end = __builtin_amdgcn_s_memtime(), r_end = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && gw < 8192) {
            nbody_sym_stamps[gw * 3 + 0] = t_end - t_beg;
            nbody_sym_stamps[gw * 3 + 1] = r_end - r_beg;
            // chunk count | HW_ID (wave slot, SIMD, CU, SE) | XCC_ID: where the dispatcher put this wave
            const unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
            const unsigned xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
            nbody_sym_stamps[gw * 3 + 2] = (unsigned long long)(k1 - k0) | ((unsigned long long)hw << 16) | ((unsigned long long)(xcc & 0xF) << 48);
        }
    }
#endif
    if (res_combine) {
        // K is a multiple of WPB: the workgroup's waves are WPB consecutive slices of ONE set; their
        // resident-side sums are added in slice order through LDS and leave as one plane row set
        __shared__ float comb[WPB - 1][3 * IPT][64];
        const int first_part = (part / WPB) * WPB;
        const int rel = part - first_part;       // 0 .. WPB-1
        if (rel > 0) {
#pragma unroll
            for (int q = 0; q < IPT; ++q) {
                comb[rel - 1][3 * q + 0][lane] = axi[q];
                comb[rel - 1][3 * q + 1][lane] = ayi[q];
                comb[rel - 1][3 * q + 2][lane] = azi[q];
            }
        }
        __syncthreads();
        if (rel == 0) {
            float4* __restrict__ out = planes + size_t(sym_sets + part / WPB) * plane_stride;
#pragma unroll
            for (int q = 0; q < IPT; ++q) {
                float sx = axi[q], sy = ayi[q], sz = azi[q];
                for (int w = 0; w < WPB - 1; ++w) { sx += comb[w][3 * q + 0][lane]; sy += comb[w][3 * q + 1][lane]; sz += comb[w][3 * q + 2][lane]; }
                out[size_t(a * IPT + q) * 64 + lane] = make_float4(sx, sy, sz, 0.f);
            }
        }
        return;
    }
    // resident side: one plane per slice index
    float4* __restrict__ out = planes + size_t(sym_sets + part) * plane_stride;
#pragma unroll
    for (int q = 0; q < IPT; ++q) out[size_t(a * IPT + q) * 64 + lane] = make_float4(axi[q], ayi[q], azi[q], 0.f);
}

// One-sided companion for sharded runs: the bodies of OTHER shards exert forces on the own bodies
// but their accelerations belong to their own GPU, so nothing is gained by updating both sides.
// Same skeleton as k_bf_sym -- a resident set of 64*IPT own bodies per wave in registers, CU-sized
// workgroups, stage-ordered evaluation -- but the partners are a contiguous slice of the
// concatenated remote segments, staged 64 at a time in the wave's private LDS tile and read back
// as wave-uniform broadcasts (ds_read_b128), 12 VALU + 1 v_rsq_f32 per interaction.  Slices are cut
// to the body (not to the chunk), so all waves carry the same number of partners.  Output: one
// resident-side plane per slice (added by k_bf_sym_reduce with the others).
template <int IPT, int WPB, bool PK>
__global__ __launch_bounds__(WPB * 64) void k_bf_os(const float4* __restrict__ pos_all,
                                                    const int* __restrict__ seg_count, int n_seg, int seg_cap,
                                                    int my_seg, int A, int K, float4* __restrict__ planes,
                                                    size_t plane_stride, float eps2) {
    __shared__ float4 stage[WPB][2][64];
    const int lane = threadIdx.x & 63;
    const int wslot = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * WPB + (wslot & 3) * (WPB / 4) + (wslot >> 2);
    if (gw >= A * K) return;
    const int a = gw / K, part = gw - a * K;
    const int n_own = seg_count[my_seg];
    const float4* __restrict__ own = pos_all + size_t(my_seg) * seg_cap;
    float eps2v = eps2;
    asm volatile("" : "+v"(eps2v));

    float xi[IPT], yi[IPT], zi[IPT], mi[IPT], axi[IPT], ayi[IPT], azi[IPT];
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
        const int i = (a * IPT + q) * 64 + lane;
        const float4 p = (i < n_own) ? own[i] : make_float4(PAD_POS, PAD_POS, PAD_POS, 0.f);
        xi[q] = p.x; yi[q] = p.y; zi[q] = p.z; mi[q] = p.w;
        axi[q] = ayi[q] = azi[q] = 0.f;
    }
    // this wave's slice [r0, r1) of the remote bodies (all segments but the own one, in order)
    long long R = 0;
    for (int s = 0; s < n_seg; ++s) if (s != my_seg) R += seg_count[s];
    const long long r0 = R * part / K, r1 = R * (part + 1) / K;
    long long seg_first = 0;  // remote index of the current segment's first body
    float dummy_x = 0.f, dummy_y = 0.f, dummy_z = 0.f;  // pair_evals' travelling-side accumulators (unused)
    v2f dummy2 = {0.f, 0.f};
    v2f xi2[IPT / 2], yi2[IPT / 2], zi2[IPT / 2], mi2[IPT / 2], axi2[IPT / 2], ayi2[IPT / 2], azi2[IPT / 2];
    if (PK) {
        pack_pairs<IPT>(xi, xi2); pack_pairs<IPT>(yi, yi2); pack_pairs<IPT>(zi, zi2); pack_pairs<IPT>(mi, mi2);
        pack_pairs<IPT>(axi, axi2); pack_pairs<IPT>(ayi, ayi2); pack_pairs<IPT>(azi, azi2);
    }
    for (int s = 0; s < n_seg; ++s) {
        if (s == my_seg) continue;
        const int ns = seg_count[s];
        const long long lo = r0 > seg_first ? r0 : seg_first;
        const long long hi = r1 < seg_first + ns ? r1 : seg_first + ns;
        const float4* __restrict__ ps = pos_all + size_t(s) * seg_cap - seg_first;  // indexable by remote index
        int buf = 0;
        if (lo < hi) {
            const long long j = lo + lane;
            stage[wslot][0][lane] = (j < hi) ? ps[j] : make_float4(PAD_POS, PAD_POS, PAD_POS, 0.f);
        }
        for (long long c0 = lo; c0 < hi; c0 += 64) {
            const long long jn = c0 + 64 + lane;  // next tile, fetched while this one is consumed
            float4 nxt = make_float4(PAD_POS, PAD_POS, PAD_POS, 0.f);
            if (c0 + 64 < hi && jn < hi) nxt = ps[jn];
            const int cnt = int(hi - c0 < 64 ? hi - c0 : 64);
            // priority = quarter of the slice still to do (see k_bf_sym): the waves of a SIMD finish together
            switch (int((4 * (r1 - c0) - 1) / (r1 - r0))) {
                case 3: __builtin_amdgcn_s_setprio(3); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                default: __builtin_amdgcn_s_setprio(0); break;
            }
            float4 pj = stage[wslot][buf][0];  // wave-uniform address: LDS broadcast
            for (int t = 0; t < cnt; ++t) {
                const float4 pn = stage[wslot][buf][(t + 1) & 63];  // the next partner lands behind this one's math
                if (PK)
                    pair_evals_pk<IPT / 2, 0, false>(xi2, yi2, zi2, mi2, axi2, ayi2, azi2, pj.x, pj.y, pj.z, pj.w, dummy2,
                                                     dummy2, dummy2, eps2v);
                else
                    pair_evals<IPT, 0, false>(xi, yi, zi, mi, axi, ayi, azi, pj.x, pj.y, pj.z, pj.w, dummy_x, dummy_y,
                                              dummy_z, eps2v);
                pj = pn;
            }
            buf ^= 1;
            stage[wslot][buf][lane] = nxt;  // same wave writes and reads its tile: program order suffices
        }
        seg_first += ns;
    }
    if (PK) { unpack_pairs<IPT>(axi2, axi); unpack_pairs<IPT>(ayi2, ayi); unpack_pairs<IPT>(azi2, azi); }
    float4* __restrict__ out = planes + size_t(part) * plane_stride;
#pragma unroll
    for (int q = 0; q < IPT; ++q) out[size_t(a * IPT + q) * 64 + lane] = make_float4(axi[q], ayi[q], azi[q], 0.f);
}

// The pairs the rotation scheme leaves out: every body against its own set (self excluded) and,
// when A is even, against the opposite set -- one-sided, since the opposite set does the same for
// ours.  64 bodies per workgroup, the 1 or 2 windows of 64*IPT partners split over 8 waves whose
// partner index is wave-uniform (scalar loads); ~3 % of the pair evaluations.
// One wave's share of those pairs: 64 bodies (one per lane) against slice `slice` of `n_slices` of
// each of the 1 or 2 windows; the caller adds the slices up.
template <int IPT>
__device__ __forceinline__ void sym_rest_wave(const float4* __restrict__ pos, int n, int A, int group, int slice,
                                              int n_slices, float& ax, float& ay, float& az, float eps2) {
    constexpr int SET = 64 * IPT;
    const int lane = threadIdx.x & 63;
    const int i = group * 64 + lane;
    const int a = (group * 64) / SET;
    const float4 pi = (i < n) ? pos[i] : make_float4(PAD_POS, PAD_POS, PAD_POS, 0.f);
    ax = ay = az = 0.f;
    const int n_win = (A % 2 == 0 && A > 1) ? 2 : 1;
    for (int w = 0; w < n_win; ++w) {
        int set = (w == 0) ? a : a + A / 2;
        if (set >= A) set -= A;
        const int j0 = set * SET + SET * slice / n_slices;
        const int j1 = min(n, set * SET + SET * (slice + 1) / n_slices);
#pragma unroll 8
        for (int j = j0; j < j1; ++j) {
            const float4 pj = pos[j];  // wave-uniform address: scalar loads
            const float dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;
            const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, eps2)));
            const float rinv = __builtin_amdgcn_rsqf(r2);
            float sj = pj.w * ((rinv * rinv) * rinv);
            sj = (j == i) ? 0.f : sj;
            ax = __builtin_fmaf(dx, sj, ax);
            ay = __builtin_fmaf(dy, sj, ay);
            az = __builtin_fmaf(dz, sj, az);
        }
    }
}

// a workgroup of NW waves: 64 bodies, the windows cut into NW slices, slices added in wave order
template <int IPT, int NW>
__device__ __forceinline__ void sym_rest_block(const float4* __restrict__ pos, int n, int A, int group,
                                               float4* __restrict__ plane, float eps2, float (*red)[3][64]) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float ax, ay, az;
    sym_rest_wave<IPT>(pos, n, A, group, wv, NW, ax, ay, az, eps2);
    if (wv > 0) { red[wv - 1][0][lane] = ax; red[wv - 1][1][lane] = ay; red[wv - 1][2][lane] = az; }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) { ax += red[w][0][lane]; ay += red[w][1][lane]; az += red[w][2][lane]; }
        plane[group * 64 + lane] = make_float4(ax, ay, az, 0.f);
    }
}

// stand-alone launch of the above (used when there is no rotation kernel to ride on)
template <int IPT>
__global__ __launch_bounds__(512) void k_bf_sym_rest(const float4* __restrict__ pos, const int* __restrict__ count,
                                                     int A, float4* __restrict__ plane, float eps2) {
    __shared__ float red[7][3][64];
    sym_rest_block<IPT, 8>(pos, *count, A, blockIdx.x, plane, eps2, red);
}

// the fixed-order sum of the planes; optionally followed at once by integrate_after_force
// (shared.rs:141-148: v += a*dt; x += (v*0.5)*dt) so that a step needs no separate K3 launch
template <bool KICK>
__global__ __launch_bounds__(256) void k_bf_sym_reduce(const float4* __restrict__ planes, int n_planes,
                                                       size_t plane_stride, const int* __restrict__ count, float g,
                                                       float4* __restrict__ acc, float4* __restrict__ pos,
                                                       float4* __restrict__ vel, float dt, const int* __restrict__ seg_count,
                                                       int n_seg, unsigned long long* __restrict__ inter) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (inter && i == 0) {   // NbodyStats::interactions of this force pass, from the live counts
        long long tot = 0;
        for (int s = 0; s < n_seg; ++s) tot += seg_count[s];
        if (tot > 0) atomicAdd(inter, (unsigned long long)(*count) * (unsigned long long)(tot - 1));
    }
    if (i >= *count) return;
    // compensated (Kahan) sum over the planes, fixed order: the kernel is bandwidth-bound, the extra
    // flops are free, and with ~1000 planes at N = 2^20 a plain f32 sum would dominate the error
    float sx = 0.f, sy = 0.f, sz = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
    for (int p = 0; p < n_planes; ++p) {
        const float4 v = planes[size_t(p) * plane_stride + i];
        const float yx = v.x - cx, yy = v.y - cy, yz = v.z - cz;
        const float tx = sx + yx, ty = sy + yy, tz = sz + yz;
        cx = (tx - sx) - yx; cy = (ty - sy) - yy; cz = (tz - sz) - yz;
        sx = tx; sy = ty; sz = tz;
    }
    const float4 a = make_float4(g * sx, g * sy, g * sz, 0.f);
    acc[i] = a;
    if (KICK) {
        float4 p = pos[i], v = vel[i];
        v.x += a.x * dt; v.y += a.y * dt; v.z += a.z * dt;
        p.x += (v.x * 0.5f) * dt; p.y += (v.y * 0.5f) * dt; p.z += (v.z * 0.5f) * dt;
        vel[i] = v;
        pos[i] = p;
    }
}

// The same with Q waves per 64 bodies, each adding up a contiguous run of the planes (fixed split, fixed order: still
// deterministic); wave 0 adds the Q partial sums.  One thread per body left a shard of 8 192 bodies with 32 workgroups
// walking ~100 planes each: 44 us, the second-longest kernel of an 8-GPU step.
template <bool KICK, int Q>
__global__ __launch_bounds__(64 * Q) void k_bf_sym_reduce_split(const float4* __restrict__ planes, int n_planes, size_t plane_stride,
                                                                const int* __restrict__ count, float g, float4* __restrict__ acc,
                                                                float4* __restrict__ pos, float4* __restrict__ vel, float dt,
                                                                const int* __restrict__ seg_count, int n_seg,
                                                                unsigned long long* __restrict__ inter) {
    __shared__ float part[Q][3][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    if (inter && blockIdx.x == 0 && threadIdx.x == 0) {   // NbodyStats::interactions of this force pass, from the live counts
        long long tot = 0;
        for (int s = 0; s < n_seg; ++s) tot += seg_count[s];
        if (tot > 0) atomicAdd(inter, (unsigned long long)(*count) * (unsigned long long)(tot - 1));
    }
    const int n = *count;
    float sx = 0.f, sy = 0.f, sz = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
    if (i < n) {
        const int p0 = int((long long)n_planes * q / Q), p1 = int((long long)n_planes * (q + 1) / Q);
        for (int p = p0; p < p1; ++p) {
            const float4 v = planes[size_t(p) * plane_stride + i];
            const float yx = v.x - cx, yy = v.y - cy, yz = v.z - cz;
            const float tx = sx + yx, ty = sy + yy, tz = sz + yz;
            cx = (tx - sx) - yx; cy = (ty - sy) - yy; cz = (tz - sz) - yz;
            sx = tx; sy = ty; sz = tz;
        }
    }
    part[q][0][lane] = sx; part[q][1][lane] = sy; part[q][2][lane] = sz;
    __syncthreads();
    if (q != 0 || i >= n) return;
    sx = sy = sz = cx = cy = cz = 0.f;
    for (int w = 0; w < Q; ++w) {
        const float yx = part[w][0][lane] - cx, yy = part[w][1][lane] - cy, yz = part[w][2][lane] - cz;
        const float tx = sx + yx, ty = sy + yy, tz = sz + yz;
        cx = (tx - sx) - yx; cy = (ty - sy) - yy; cz = (tz - sz) - yz;
        sx = tx; sy = ty; sz = tz;
    }
    const float4 a = make_float4(g * sx, g * sy, g * sz, 0.f);
    acc[i] = a;
    if (KICK) {
        float4 p = pos[i], v = vel[i];
        v.x += a.x * dt; v.y += a.y * dt; v.z += a.z * dt;
        p.x += (v.x * 0.5f) * dt; p.y += (v.y * 0.5f) * dt; p.z += (v.z * 0.5f) * dt;
        vel[i] = v;
        pos[i] = p;
    }
}

}  // namespace nbody
#ifdef NBODY_TUNING
namespace nbody { int read_sym_stamps(unsigned long long* out, int n_waves); }
extern "C" int nbody_sym_read_stamps(unsigned long long* out, int n_waves) { return nbody::read_sym_stamps(out, n_waves); }
#endif
namespace nbody {

// ----------------------------------------------------------------------------------- host side
// 8 bodies per lane of a resident set; 4 for small shards (up to 10 240 bodies): twice the waves with half the work each --
// 4 096 bodies 22.0 -> 15.5 us, 8 192 (a rank's shard of the metric at 8 GPUs) 25.9 -> 23.4 us; from 12 288 up 8 is ahead
int sym_bodies_per_lane(size_t n) {
    const Tuning& t = tuning();
    if (!t.sym_packed || t.sym_wpb != 4 || t.sym_ipt == 8) return 8;
    if (t.sym_ipt == 4) return 4;
    return n <= 10240 ? 4 : 8;
}

SymPlan make_sym_plan(int n_upper) {
    SymPlan p;
    const int IPT = sym_bodies_per_lane(size_t(n_upper));   // bodies per lane of a resident set
    p.ipt = IPT;
    p.A = (n_upper + 64 * IPT - 1) / (64 * IPT);
    p.sym_sets = (p.A + 1) / 2 - 1;             // ceil(A/2) - 1 sets are met symmetrically
    const int L = IPT * p.sym_sets;              // chunk visits per set, all of equal cost
    // Workgroups of 4 waves (one per SIMD; the register budget admits three per CU) unless tuned otherwise.  Round 2 used
    // CU-sized workgroups of 12: at mid sizes their grid is smaller than the chip (N = 16 384: 160 workgroups on 256 CUs,
    // 62 % of the rate the kernel reaches at N = 65 536) and even at N = 65 536 three small workgroups per CU finish 2.8 %
    // earlier (profiles/r03_sym_plan_sweep.txt: K and workgroup size swept at 12 sizes).
    const int want = tuning().sym_wpb;
    p.wpb = want == 8 ? 8 : want == 12 ? 12 : want == 16 ? 16 : 4;
    int K = 1;
    if (p.wpb == 4) {
        // waves per set, from that sweep: one-chunk slices while the set's chunk sequence is short (every wave then has the
        // same work and there are at most ~2 waves per SIMD), two-chunk slices up to L = 160, beyond that 64 slices per set
        // (a multiple of the workgroup size, so the resident-side sums of a workgroup leave as one plane)
        K = L <= 100 ? L : L <= 160 ? (L + 1) / 2 : 64;
    } else {
        const int resident_wgs_per_cu = (p.wpb == 8) ? 2 : 1;
        const int slots = 256 * resident_wgs_per_cu * p.wpb;  // waves the chip holds at once
        // A * K waves run in ceil(A*K/slots) rounds; pick the K (few, long slices preferred) whose last
        // round is fullest -- e.g. A = 2048 sets: K = 1 would fill 2/3 of one round, K = 3 fills two
        double best = 0.0;
        const int k_hi = std::min(126, std::min(L, std::max(1, slots * std::max(1, tuning().sym_rounds) * 4 / p.A)));
        for (int k = 1; k <= k_hi; ++k) {
            const long long waves = (long long)p.A * k;
            const long long rounds = (waves + slots - 1) / slots;
            if (rounds > 4 * std::max(1, tuning().sym_rounds) && k > 1) break;
            const double fill = double(waves) / double(rounds * slots);
            const double even = (L % k == 0) ? 1.0 : double(L / k) / double(L / k + 1);  // shortest/longest slice
            const double per_wave = double(L) / k / (double(L) / k + 0.3);  // set-up cost of a wave ~ 0.3 chunk
            if (fill * even * per_wave > best + 1e-9) { best = fill * even * per_wave; K = k; }
        }
    }
    if (tuning().sym_k > 0) K = std::min(126, tuning().sym_k);   // (the cut points live in a 128-entry device array)
    if (K > L) K = L;
    if (K < 1) K = 1;
    p.K = K;
    p.res_combine = (K % p.wpb == 0) ? 1 : 0;    // a workgroup = wpb slices of one set: one resident plane per workgroup
    p.k_res = p.res_combine ? K / p.wpb : K;
    p.n_planes = p.sym_sets + p.k_res + 1;       // travelling-side, resident-side, own/opposite set
    p.n_pad = size_t(p.A) * 64 * IPT;
    p.plane_stride = p.n_pad;
    p.bounds.resize(K + 1);
    for (int part = 0; part <= K; ++part) p.bounds[part] = int((long long)L * part / K);
    return p;
}

#ifdef NBODY_TUNING
int read_sym_stamps(unsigned long long* out, int n_waves) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(nbody_sym_stamps), sizeof(unsigned long long) * 3 * n_waves) == hipSuccess ? 0 : -1;
}
#endif

uint64_t sym_main_pairs(const SymPlan& p, size_t n) {
    const size_t set = size_t(64) * p.ipt;
    auto size_of = [&](int a) { size_t lo = size_t(a) * set; return lo >= n ? size_t(0) : std::min(set, n - lo); };
    uint64_t pairs = 0;
    for (int a = 0; a < p.A; ++a)
        for (int d = 1; d <= p.sym_sets; ++d) pairs += uint64_t(size_of(a)) * uint64_t(size_of((a + d) % p.A));
    return pairs;
}

// the rotation kernel alone (the dominant launch: bench.py times it with HIP events)
void launch_bf_sym_main(hipStream_t s, const Shard& sh, const SymPlan& p, const int* d_bounds, float4* planes,
                        int n_upper, float g_soft2) {
    if (n_upper <= 0 || p.sym_sets <= 0) return;
    const int main_blocks = (p.A * p.K + p.wpb - 1) / p.wpb;
    const int rest_blocks = int(p.n_pad / 64);  // one workgroup per 64 bodies
    const dim3 grid(main_blocks + rest_blocks), block(p.wpb * 64);
#define SYM_LAUNCH(WPB, DBG) hipLaunchKernelGGL((k_bf_sym<8, WPB, DBG, PKV>), grid, block, 0, s, sh.own_pos(), sh.own_count(), p.A, p.K, d_bounds, p.sym_sets, planes, p.plane_stride, g_soft2, p.res_combine)
#ifdef NBODY_TUNING   // in-kernel stamps (tools/sym_cycles.py: 4) and timing experiments that do not compute the forces (5-7)
    const int dbg = tuning().sym_debug;
#else
    constexpr int dbg = 0;
#endif
    if (p.ipt == 4) {   // half-size resident sets (4 bodies per lane): twice the waves for small shards
        hipLaunchKernelGGL((k_bf_sym<4, 4, 0, true>), grid, block, 0, s, sh.own_pos(), sh.own_count(), p.A, p.K, d_bounds, p.sym_sets, planes, p.plane_stride, g_soft2, p.res_combine);
        return;
    }
    if (tuning().sym_packed) {
        constexpr bool PKV = true;
        if (dbg == 4) {
#ifdef NBODY_TUNING
            if (p.wpb == 12) SYM_LAUNCH(12, 4); else if (p.wpb == 8) SYM_LAUNCH(8, 4); else if (p.wpb == 4) SYM_LAUNCH(4, 4); else SYM_LAUNCH(16, 4);
#endif
        } else {
            if (p.wpb == 12) SYM_LAUNCH(12, 0); else if (p.wpb == 8) SYM_LAUNCH(8, 0); else if (p.wpb == 4) SYM_LAUNCH(4, 0); else SYM_LAUNCH(16, 0);
        }
    } else {
        constexpr bool PKV = false;
        if (dbg >= 4 && dbg <= 7) {
#ifdef NBODY_TUNING
            if (dbg == 4) { if (p.wpb == 12) SYM_LAUNCH(12, 4); else if (p.wpb == 8) SYM_LAUNCH(8, 4); else if (p.wpb == 4) SYM_LAUNCH(4, 4); else SYM_LAUNCH(16, 4); }
            else if (dbg == 5) SYM_LAUNCH(16, 5);
            else if (dbg == 6) SYM_LAUNCH(16, 6);
            else SYM_LAUNCH(16, 7);
#endif
        } else {
            if (p.wpb == 12) SYM_LAUNCH(12, 0); else if (p.wpb == 8) SYM_LAUNCH(8, 0); else if (p.wpb == 4) SYM_LAUNCH(4, 0); else SYM_LAUNCH(16, 0);
        }
    }
#undef SYM_LAUNCH
}

// one-sided forces of the other shards' bodies on the own bodies: K_os planes starting at `planes`
void launch_bf_os(hipStream_t s, const Shard& sh, int A, int K, float4* planes, size_t plane_stride, float g_soft2) {
    if (A <= 0 || K <= 0) return;
    const int wpb = 12;
    if (tuning().sym_packed)
        hipLaunchKernelGGL((k_bf_os<8, 12, true>), dim3((A * K + wpb - 1) / wpb), dim3(wpb * 64), 0, s, sh.pos_all,
                           sh.seg_count, sh.n_seg, sh.seg_cap, sh.my_seg, A, K, planes, plane_stride, g_soft2);
    else
        hipLaunchKernelGGL((k_bf_os<8, 12, false>), dim3((A * K + wpb - 1) / wpb), dim3(wpb * 64), 0, s, sh.pos_all,
                           sh.seg_count, sh.n_seg, sh.seg_cap, sh.my_seg, A, K, planes, plane_stride, g_soft2);
}

// the fixed-order sum of the planes into acc (the own/opposite-set pairs ride on the rotation
// kernel's launch; without one they get their own).  kick_dt != nullptr fuses integrate_after_force.
void launch_bf_sym_tail(hipStream_t s, const Shard& sh, const SymPlan& p, float4* planes, int n_upper, float g,
                        float g_soft2, const float* kick_dt) {
    if (n_upper <= 0) return;
    if (p.sym_sets == 0) {  // no rotation pass: the resident-side planes are never written
        float4* resident0 = planes + size_t(p.sym_sets) * p.plane_stride;
        (void)hipMemsetAsync(resident0, 0, size_t(p.k_res) * p.plane_stride * sizeof(float4), s);
        if (p.ipt == 4)
            hipLaunchKernelGGL(k_bf_sym_rest<4>, dim3(int(p.n_pad / 64)), dim3(512), 0, s, sh.own_pos(),
                               sh.own_count(), p.A, planes + size_t(p.sym_sets + p.k_res) * p.plane_stride, g_soft2);
        else
        hipLaunchKernelGGL(k_bf_sym_rest<8>, dim3(int(p.n_pad / 64)), dim3(512), 0, s, sh.own_pos(),
                           sh.own_count(), p.A, planes + size_t(p.sym_sets + p.k_res) * p.plane_stride, g_soft2);
    }
    const float dt = kick_dt ? *kick_dt : 0.f;
#define REDUCE_ARGS planes, p.n_planes, p.plane_stride, sh.own_count(), g, sh.acc, sh.own_pos(), sh.vel, dt, sh.seg_count, sh.n_seg, sh.inter
    if (tuning().sym_reduce_split && p.n_planes >= 16) {   // several waves per 64 bodies: 16 for small shards, 4 otherwise
        const dim3 grid((n_upper + 63) / 64);
        if (n_upper <= 16384) {
            if (kick_dt) hipLaunchKernelGGL((k_bf_sym_reduce_split<true, 16>), grid, dim3(1024), 0, s, REDUCE_ARGS);
            else hipLaunchKernelGGL((k_bf_sym_reduce_split<false, 16>), grid, dim3(1024), 0, s, REDUCE_ARGS);
        } else {
            if (kick_dt) hipLaunchKernelGGL((k_bf_sym_reduce_split<true, 4>), grid, dim3(256), 0, s, REDUCE_ARGS);
            else hipLaunchKernelGGL((k_bf_sym_reduce_split<false, 4>), grid, dim3(256), 0, s, REDUCE_ARGS);
        }
    } else {
        const dim3 grid((n_upper + 255) / 256);
        if (kick_dt) hipLaunchKernelGGL(k_bf_sym_reduce<true>, grid, dim3(256), 0, s, REDUCE_ARGS);
        else hipLaunchKernelGGL(k_bf_sym_reduce<false>, grid, dim3(256), 0, s, REDUCE_ARGS);
    }
#undef REDUCE_ARGS
}

}  // namespace nbody
