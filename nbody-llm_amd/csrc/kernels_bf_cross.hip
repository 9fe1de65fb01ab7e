// kernels_bf_cross.hip -- the symmetric all-pairs scheme ACROSS shards (multi-GPU, fast math).
//
// With G shards the pair space between different shards is dealt so that every unordered pair
// {body of shard r, body of shard q} is evaluated once, by one of the two GPUs, which updates both
// bodies: rank r is "resident" for the partners r+1 .. r+ceil(G/2)-1 (cyclic); for even G the pair
// of opposite ranks (r, r+G/2) is split down the middle -- the lower rank keeps all its bodies
// resident against the first half of the higher rank's chunks, the higher rank keeps the second
// half of its own sets resident against all of the lower rank's chunks.  The partial sums a GPU
// accumulates for another shard's bodies travel back to their owner in one grouped
// ncclSend/ncclRecv round per step (nbody_api.cpp); the receiver adds them as further planes, in
// rank-distance order: deterministic.
//
// The kernel is k_bf_sym's rotation scheme (resident set of 64*IPT bodies in registers,
// travelling chunks of 64 rotated through the lanes with ds_bpermute, 16 VALU + 1 v_rsq_f32 per
// unordered pair) with the chunk sequence of a resident set made of the partner shards' chunks.
// Work is handed out as host-built slices {set, first chunk, last chunk, plane}: sets that take
// part in more partners get more slices.
#include "kernels.h"
#include "bf_pair.h"

#include <algorithm>

// tuning knobs of the plan (tools/tune_sharded.py); every rank of a world must use the same values
                                          // 12 = one workgroup per CU left CUs idle: G = 8, N = 65 536: 0.100 against 0.114 ms)

namespace nbody {

template <int IPT, int WPB, bool PK>
__global__ __launch_bounds__(WPB * 64) void k_bf_cross(const float4* __restrict__ pos_all,
                                                       const int* __restrict__ seg_count, int seg_cap, int my_seg,
                                                       CrossPartners parts, const int4* __restrict__ slices,
                                                       int n_slices, int A, float4* __restrict__ res_planes,
                                                       float4* __restrict__ xplanes, size_t plane_stride,
                                                       float eps2) {
    const int lane = threadIdx.x & 63;
    const int wslot = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * WPB + (wslot & 3) * (WPB / 4) + (wslot >> 2);
    if (gw >= n_slices) return;
    const int4 sl = slices[gw];  // {set, k0, k1, resident plane}
    const int a = sl.x, k0 = sl.y, k1 = sl.z;
    const int n_own = seg_count[my_seg];
    const float4* __restrict__ own = pos_all + size_t(my_seg) * seg_cap;
    const int src_lane = ((lane + 63) & 63) * 4;
    const int src_lane2 = ((lane + 62) & 63) * 4;
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

    // position k of this set's chunk sequence -> (partner, chunk): the partners this set takes part
    // in, one after the other
    int part = 0, part_first = 0;  // sequence index of the current partner's first chunk
    auto locate = [&](int k, int& pi, int& c) {
        while (true) {
            const bool mine = a >= parts.a0[part] && a < parts.a1[part];
            const int len = mine ? parts.c1[part] - parts.c0[part] : 0;
            if (k < part_first + len) break;
            part_first += len;
            ++part;
        }
        pi = part;
        c = parts.c0[part] + (k - part_first);
    };
    auto fetch = [&](int k, int& pi, int& c, int& ns) {  // chunk k of the sequence: where it is and its bodies
        locate(k, pi, c);
        const int s = parts.seg[pi];
        ns = seg_count[s];
        const int j = c * 64 + lane;
        return (j < ns) ? pos_all[size_t(s) * seg_cap + j] : make_float4(PAD_POS, PAD_POS, PAD_POS, 0.f);
    };
    // packed form (PK): the resident bodies two per register pair, see bf_pair.h
    v2f xi2[IPT / 2], yi2[IPT / 2], zi2[IPT / 2], mi2[IPT / 2], axi2[IPT / 2], ayi2[IPT / 2], azi2[IPT / 2];
    if (PK) {
        pack_pairs<IPT>(xi, xi2); pack_pairs<IPT>(yi, yi2); pack_pairs<IPT>(zi, zi2); pack_pairs<IPT>(mi, mi2);
        pack_pairs<IPT>(axi, axi2); pack_pairs<IPT>(ayi, ayi2); pack_pairs<IPT>(azi, azi2);
    }
    int pi = 0, c = 0, ns = 0, pi_n = 0, c_n = 0, ns_n = 0;
    float4 nxt = (k0 < k1) ? fetch(k0, pi_n, c_n, ns_n) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = k0; k < k1; ++k) {
        const float4 pj = nxt;
        pi = pi_n; c = c_n; ns = ns_n;
        if (k + 1 < k1) nxt = fetch(k + 1, pi_n, c_n, ns_n);  // lands behind this chunk's 64 steps
        if (c * 64 >= ns) {  // a chunk beyond the partner's live bodies: its partial sums are zero
            xplanes[(size_t(pi) * A + a) * plane_stride + size_t(c) * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        float xj = pj.x, yj = pj.y, zj = pj.z, mj = pj.w;
        float axj = 0.f, ayj = 0.f, azj = 0.f;
        v2f axj2 = {0.f, 0.f}, ayj2 = {0.f, 0.f}, azj2 = {0.f, 0.f};
        switch ((4 * (k1 - k) - 1) / (k1 - k0)) {  // progress-based priority, see k_bf_sym
            case 3: __builtin_amdgcn_s_setprio(3); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(0); break;
        }
        if (PK) {  // one step ahead, two steps per iteration: the two register sets swap roles (see k_bf_sym)
            for (int st = 0; st < 64; st += 2) {
                const float x1 = rotl(xj, src_lane), y1 = rotl(yj, src_lane), z1 = rotl(zj, src_lane), m1 = rotl(mj, src_lane);
                pair_evals_pk<IPT / 2, 0, true>(xi2, yi2, zi2, mi2, axi2, ayi2, azi2, xj, yj, zj, mj, axj2, ayj2, azj2, eps2v);
                axj2.x = rotl(axj2.x, src_lane); ayj2.x = rotl(ayj2.x, src_lane); azj2.x = rotl(azj2.x, src_lane);
                axj2.y = rotl(axj2.y, src_lane); ayj2.y = rotl(ayj2.y, src_lane); azj2.y = rotl(azj2.y, src_lane);
                xj = rotl(x1, src_lane); yj = rotl(y1, src_lane); zj = rotl(z1, src_lane); mj = rotl(m1, src_lane);
                pair_evals_pk<IPT / 2, 0, true>(xi2, yi2, zi2, mi2, axi2, ayi2, azi2, x1, y1, z1, m1, axj2, ayj2, azj2, eps2v);
                axj2.x = rotl(axj2.x, src_lane); ayj2.x = rotl(ayj2.x, src_lane); azj2.x = rotl(azj2.x, src_lane);
                axj2.y = rotl(axj2.y, src_lane); ayj2.y = rotl(ayj2.y, src_lane); azj2.y = rotl(azj2.y, src_lane);
            }
        } else {
            float x1 = rotl(xj, src_lane), y1 = rotl(yj, src_lane), z1 = rotl(zj, src_lane), m1 = rotl(mj, src_lane);
#pragma unroll 2
            for (int st = 0; st < 64; ++st) {
                const float x2 = rotl(xj, src_lane2), y2 = rotl(yj, src_lane2), z2 = rotl(zj, src_lane2), m2 = rotl(mj, src_lane2);
                pair_evals<IPT, 0, true>(xi, yi, zi, mi, axi, ayi, azi, xj, yj, zj, mj, axj, ayj, azj, eps2v);
                axj = rotl(axj, src_lane); ayj = rotl(ayj, src_lane); azj = rotl(azj, src_lane);
                xj = x1; yj = y1; zj = z1; mj = m1;
                x1 = x2; y1 = y2; z1 = z2; m1 = m2;
            }
        }
        if (PK) { axj = axj2.x + axj2.y; ayj = ayj2.x + ayj2.y; azj = azj2.x + azj2.y; }
        xplanes[(size_t(pi) * A + a) * plane_stride + size_t(c) * 64 + lane] = make_float4(axj, ayj, azj, 0.f);
    }
    if (PK) { unpack_pairs<IPT>(axi2, axi); unpack_pairs<IPT>(ayi2, ayi); unpack_pairs<IPT>(azi2, azi); }
    // resident side (zeros for the padding slices k0 == k1 that complete a set's plane count)
    float4* __restrict__ out = res_planes + size_t(sl.w) * plane_stride;
#pragma unroll
    for (int q = 0; q < IPT; ++q) out[size_t(a * IPT + q) * 64 + lane] = make_float4(axi[q], ayi[q], azi[q], 0.f);
}

// partial sums for partner `pi`'s bodies: the planes of the resident sets that met them, added in
// set order (compensated) into the buffer that is sent to the owner
__global__ __launch_bounds__(256) void k_bf_cross_reduce(const float4* __restrict__ xplanes, size_t plane_stride,
                                                         CrossPartners parts, int A, int seg_cap,
                                                         float4* __restrict__ send) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int pi = blockIdx.y;
    if (j >= seg_cap) return;
    float sx = 0.f, sy = 0.f, sz = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
    const int c = j >> 6;
    if (c >= parts.c0[pi] && c < parts.c1[pi]) {
        for (int a = parts.a0[pi]; a < parts.a1[pi]; ++a) {
            const float4 v = xplanes[(size_t(pi) * A + a) * plane_stride + j];
            const float yx = v.x - cx, yy = v.y - cy, yz = v.z - cz;
            const float tx = sx + yx, ty = sy + yy, tz = sz + yz;
            cx = (tx - sx) - yx; cy = (ty - sy) - yy; cz = (tz - sz) - yz;
            sx = tx; sy = ty; sz = tz;
        }
    }
    send[size_t(pi) * plane_stride + j] = make_float4(sx, sy, sz, 0.f);
}

// ----------------------------------------------------------------------------------- host side
CrossPlan make_cross_plan(int rank, int G, int seg_cap, int n_own_upper) {
    CrossPlan p;
    // resident sets of 8 bodies per lane; 4 when the shards are small: a chunk visit is the unit of work
    // and with few sets there are too few visits per wave to deal them evenly (fewer than one per slot; at one to three
    // per wave 8 with packed pairs is already ahead: G = 8, N = 65 536: 0.100 ms against 0.122 with 4).  Decided from the shard
    // CAPACITY, which every rank knows, so that all ranks cut the opposite-rank pair at the same chunk.
    const int chunks_cap = (seg_cap + 63) / 64;                 // chunks a shard can hold
    const long long visits8 = (long long)((seg_cap + 511) / 512) * chunks_cap * std::max(1, G / 2);
    const int IPT = (tuning().cross_ipt == 4 || tuning().cross_ipt == 8) ? tuning().cross_ipt : (visits8 < 3072) ? 4 : 8;
    p.ipt = IPT;
    p.A = (n_own_upper + 64 * IPT - 1) / (64 * IPT);
    // opposite ranks: the lower rank takes the higher rank's chunks below `split`, the higher rank keeps
    // its own sets from split/IPT on resident (a multiple of 8 chunks, so a set boundary for IPT 4 and 8;
    // NOT clipped to the capacity: when it lies beyond it the lower rank simply takes everything)
    const int split = ((chunks_cap + 7) / 8 + 1) / 2 * 8;
    const int split_chunk = std::min(chunks_cap, split);
    const int split_set = split / IPT;
    CrossPartners& q = p.parts;
    q.n = 0;
    auto add = [&](int seg, int c0, int c1, int a0, int a1) {
        // empty ranges stay in the list: the partner still expects a (zero) message from this rank
        if (q.n >= CrossPartners::kMax) return;
        c1 = std::max(c0, c1); a1 = std::max(a0, a1);
        q.seg[q.n] = seg; q.c0[q.n] = c0; q.c1[q.n] = c1; q.a0[q.n] = a0; q.a1[q.n] = a1;
        ++q.n;
    };
    const int half = (G + 1) / 2;  // ceil(G/2)
    for (int d = 1; d < half; ++d) add((rank + d) % G, 0, chunks_cap, 0, p.A);
    if (G % 2 == 0 && G >= 2) {
        const int opp = (rank + G / 2) % G;
        if (rank < opp) add(opp, 0, split_chunk, 0, p.A);                    // all own sets x their first half
        else add(opp, 0, chunks_cap, std::min(p.A, split_set), p.A);         // own second half x all of theirs
    }
    // who sends to me: the ranks I am a partner of (mirror image of the above)
    p.n_recv = 0;
    for (int d = 1; d < half; ++d) p.recv_from[p.n_recv++] = (rank - d + G) % G;
    if (G % 2 == 0 && G >= 2) p.recv_from[p.n_recv++] = (rank + G / 2) % G;
    // slices: ~equal length over all sets, at most kMaxPlanes per set, zero slices complete the count
    std::vector<int> L(p.A, 0);
    long long total = 0;
    for (int a = 0; a < p.A; ++a) {
        for (int i = 0; i < q.n; ++i) if (a >= q.a0[i] && a < q.a1[i]) L[a] += q.c1[i] - q.c0[i];
        total += L[a];
    }
    const int slots = std::max(256, tuning().cross_slots);
    const double target = std::max(1.0, double(total) / double(slots));
    int kmax = 1;
    std::vector<int> Ka(p.A, 0);
    for (int a = 0; a < p.A; ++a) {
        Ka[a] = L[a] ? std::max(1, std::min(std::min(CrossPlan::kMaxPlanes, L[a]), int(double(L[a]) / target + 0.5))) : 0;
        kmax = std::max(kmax, Ka[a]);
    }
    p.k_res = kmax;
    p.slices.clear();
    for (int pl = 0; pl < kmax; ++pl)   // real slices first, interleaved over the sets
        for (int a = 0; a < p.A; ++a)
            if (pl < Ka[a]) p.slices.push_back(make_int4(a, int((long long)L[a] * pl / Ka[a]), int((long long)L[a] * (pl + 1) / Ka[a]), pl));
    for (int pl = 0; pl < kmax; ++pl)
        for (int a = 0; a < p.A; ++a)
            if (pl >= Ka[a]) p.slices.push_back(make_int4(a, 0, 0, pl));
    return p;
}

void launch_bf_cross(hipStream_t s, const Shard& sh, const CrossPlan& p, const int4* d_slices, float4* res_planes,
                     float4* xplanes, float4* send, size_t plane_stride, float g_soft2) {
    if (p.slices.empty() || p.parts.n == 0) return;
    const int n_slices = int(p.slices.size());
#define CROSS_LAUNCH(IPT, WPB, PKV) hipLaunchKernelGGL((k_bf_cross<IPT, WPB, PKV>), dim3((n_slices + WPB - 1) / WPB), dim3(WPB * 64), 0, s, sh.pos_all, sh.seg_count, sh.seg_cap, sh.my_seg, p.parts, d_slices, n_slices, p.A, res_planes, xplanes, plane_stride, g_soft2)
#define CROSS_LAUNCH_W(IPT, PKV) do { if (tuning().cross_wpb == 4) CROSS_LAUNCH(IPT, 4, PKV); else if (tuning().cross_wpb == 8) CROSS_LAUNCH(IPT, 8, PKV); else CROSS_LAUNCH(IPT, 12, PKV); } while (0)
    // packed pairs only with 8 bodies per lane: with 4 a stage is two instructions long and the dependent
    // stages stall on each other (G = 8 at N = 65 536: 0.140 ms packed, 0.119 ms scalar)
    if (p.ipt == 4) CROSS_LAUNCH_W(4, false);
    else if (tuning().sym_packed) CROSS_LAUNCH_W(8, true);
    else CROSS_LAUNCH_W(8, false);
#undef CROSS_LAUNCH_W
#undef CROSS_LAUNCH
    hipLaunchKernelGGL(k_bf_cross_reduce, dim3((sh.seg_cap + 255) / 256, p.parts.n), dim3(256), 0, s, xplanes,
                       plane_stride, p.parts, p.A, sh.seg_cap, send);
}

}  // namespace nbody
