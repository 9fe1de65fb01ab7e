// kernels_let.hip -- Barnes-Hut over SPATIAL shards with a halo ("locally essential tree") exchange
// (BASELINE.json configs[4]: "spatial shard 8 x MI355X with halo cell exchange").
//
// The tree is the one the single-shard device build makes (kernels_tree.hip: same cells, pre-order, skip links as
// BarnesHutSimulation::build_tree, src/manual/barnes_hut.rs:143-183), but no rank ever sees all bodies:
//   * ownership by Morton-key range: rank r holds the bodies whose 63-bit key (the orthant codes of 21 levels) lies in
//     [bound[r], bound[r+1]); the bounds are the G-quantiles of the world's keys -- weighted by the visit counts of the
//     last walk, so that the ranks get equal work -- redrawn every step (k_let_contrib, k_let_offsets).  Bodies on the
//     wrong side of a bound migrate (k_let_classify / k_let_pack_migrants pack them per destination, k_let_append takes
//     them in);
//   * then the GLOBAL sorted order is the concatenation of the ranks' local sorted orders, so in the build's
//     formulation (body k opens the cells of depths delta[k-1]+1 .. delta[k]; its leaf sits at max(..)+1) only a rank's
//     FIRST and LAST sorted body have a neighbour elsewhere: two "edge" values (k_let_edges) from an all-gather of every
//     rank's first and last key, and each rank emits a contiguous slice of the global pre-order node array;
//   * a cell is opened by its first body, so a cell spanning ranks belongs to exactly one of them and lies on that
//     rank's LAST body's path: at most 21 per rank.  Its sums need the later ranks' bodies below its upper key and its
//     skip link the first node behind them: every rank works out what it adds to every earlier rank's spanning cells
//     (k_let_contrib), one all-gather of those small tables, and everybody knows all spanning cells (k_let_finalize);
//   * everything else is private to a rank's slice.  A partner needs a private node only if one of its bodies can get
//     there, i.e. if it can OPEN every ancestor: k_let_open_masks / k_let_flag_count test the ancestors against the
//     boxes the partners' bodies lie in (opening test of barnes_hut.rs:192 with the box's nearest point) and write one
//     list of records per partner,
//     one variable-size send/recv round; the receiver lays the nodes it HOLDS -- its slice and the imports -- out in the order
//     of their global indices and turns every link into a position in that order (launch_assemble).
// The walk (k_bh_walk) then runs over that array as over a complete tree: every node a body visits is there.
// Fast math, device build; node values come from f64 prefix sums over the LOCAL sorted order, so against the single-shard
// device build a centre of mass can differ in its last f32 bit (counts agree to ~1e-6, tested).
#include "kernels_let.h"

namespace nbody {
namespace let {

namespace {

struct Sum4 { double m, x, y, z; };

__device__ __forceinline__ int common_levels(unsigned long long a, unsigned long long b) {
    const unsigned long long x = a ^ b;
    if (x == 0) return kLevels;
    return (__clzll((long long)x) - 1) / 3;   // bit 63 is unused
}
__device__ __forceinline__ unsigned long long prefix_lo(unsigned long long key, int d) {   // smallest key sharing d levels
    const int sh = 3 * (kLevels - d);
    return sh >= 63 ? 0ull : (key >> sh) << sh;
}
__device__ __forceinline__ unsigned long long prefix_hi(unsigned long long key, int d) {   // largest key sharing d levels
    const int sh = 3 * (kLevels - d);
    return sh >= 63 ? 0x7fffffffffffffffull : (prefix_lo(key, d) | ((1ull << sh) - 1ull));
}
__device__ __forceinline__ unsigned long long key_of(const float4 p, float cx, float cy, float cz, float width) {
    float hw = width * 0.5f;  // Bounds::new; the recurrences of kernels_tree.hip k_tree_keys
    unsigned long long key = 0;
#pragma unroll 1
    for (int l = 0; l < kLevels; ++l) {
        const bool bx = p.x > cx, by = p.y > cy, bz = p.z > cz;  // get_orthant
        key = (key << 3) | (unsigned long long)((bx ? 1 : 0) | (by ? 2 : 0) | (bz ? 4 : 0));
        hw = hw * 0.5f;                                            // create_orthant
        cx = bx ? cx + hw : cx - hw;
        cy = by ? cy + hw : cy - hw;
        cz = bz ? cz + hw : cz - hw;
    }
    return key;
}
// order-preserving map float -> int for atomicMin / atomicMax
__device__ __forceinline__ int f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// Could some body inside the box fail the acceptance test w^2 < theta2 * r^2 on this node (i.e. open it)?  Conservative:
// "no" only when the box's nearest point accepts it with a margin that covers the rounding of the walk's own r^2.
__device__ __forceinline__ bool box_could_open(const float4 A, float w2, const float* lo, const float* hi, float theta2) {
    const float dx = fmaxf(0.f, fmaxf(lo[0] - A.x, A.x - hi[0]));
    const float dy = fmaxf(0.f, fmaxf(lo[1] - A.y, A.y - hi[1]));
    const float dz = fmaxf(0.f, fmaxf(lo[2] - A.z, A.z - hi[2]));
    const float d2 = dx * dx + dy * dy + dz * dz;
    return !(w2 < theta2 * d2 * 0.9999f);
}
// ---- migration: a body whose key left this rank's range goes to the rank that owns it.  Two passes, so that the
// emigrants of each destination end up contiguous in ONE buffer of seg_cap records whatever their number (a thin disc
// cut at z = 0 trades thousands of bodies per step between two ranks): classify + count, offsets, pack.
__global__ __launch_bounds__(256) void k_let_classify(const float4* __restrict__ pos, const int* __restrict__ count, float cx, float cy,
                                                      float cz, float width, const unsigned long long* __restrict__ bounds, int G, int me,
                                                      unsigned char* __restrict__ keep, unsigned char* __restrict__ dest_of,
                                                      int* __restrict__ escaped, int* __restrict__ send_count, int after_drift) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= *count) return;
    dest_of[k] = 255;
    if (after_drift && !keep[k]) return;   // left the box in this step's drift: the retain pass below drops it with the emigrants
    const unsigned long long key = key_of(pos[k], cx, cy, cz, width);
    int dest = 0;   // last r with bounds[r] <= key
    for (int r = 1; r < G; ++r) if (bounds[r] <= key) dest = r;
    keep[k] = dest == me ? 1 : 0;      // (outside a step every flag is written here: the old ones belong to an earlier retain)
    if (dest == me) return;            // an emigrant leaves this rank: the retain pass (k_compact) closes the gap
    dest_of[k] = (unsigned char)dest;
    atomicAdd(escaped, 1);
    atomicAdd(&send_count[dest], 1);
}
__global__ void k_let_mig_offsets(const int* __restrict__ send_count, int G, int* __restrict__ send_off, int* __restrict__ cursor) {
    if (threadIdx.x != 0) return;
    int run = 0;
    for (int r = 0; r < G; ++r) { send_off[r] = run; run += send_count[r]; cursor[r] = 0; }
    send_off[G] = run;
}
__global__ __launch_bounds__(256) void k_let_pack_migrants(const float4* __restrict__ pos, const float4* __restrict__ vel,
                                                           const float4* __restrict__ acc, const int* __restrict__ ids,
                                                           const int* __restrict__ count, const unsigned char* __restrict__ dest_of,
                                                           const int* __restrict__ send_off, int* __restrict__ cursor,
                                                           Migrant* __restrict__ send) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= *count) return;
    const int dest = dest_of[k];
    if (dest == 255) return;
    const int slot = send_off[dest] + atomicAdd(&cursor[dest], 1);   // (<= the bodies this rank holds: the buffer has seg_cap records)
    Migrant m;
    m.pos = pos[k]; m.vel = vel[k]; m.acc = acc[k]; m.id = ids[k]; m.pad[0] = m.pad[1] = m.pad[2] = 0;
    send[slot] = m;
}

// received migrants (n_in of them, from whatever ranks) behind the own bodies
__global__ __launch_bounds__(256) void k_let_append(float4* __restrict__ pos, float4* __restrict__ vel, float4* __restrict__ acc,
                                                    int* __restrict__ ids, int* __restrict__ count, int cap,
                                                    const Migrant* __restrict__ recv, int n_in, int* __restrict__ flags,
                                                    int* __restrict__ new_count) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int n0 = *count;
    if (j == 0) {
        if (n0 + n_in > cap) atomicOr(flags, kFlagCapacity);
        *new_count = min(cap, n0 + n_in);   // (applied by k_let_commit_count: every thread of this launch reads the old count)
    }
    if (j >= n_in) return;
    const int d = n0 + j;
    if (d >= cap) return;
    const Migrant m = recv[j];
    pos[d] = m.pos; vel[d] = m.vel; acc[d] = m.acc; ids[d] = m.id;
}
__global__ void k_let_commit_count(int* __restrict__ count, const int* __restrict__ new_count, int* __restrict__ send_count, int G,
                                   int* __restrict__ migrated, const int* __restrict__ flags) {
    if (flags && (*flags & kFlagMigSpill)) return;   // the migrant round did not fit its posted sizes: it is made again, nothing is committed
    if (threadIdx.x == 0) *count = *new_count;
    if (int(threadIdx.x) < G) {
        if (send_count[threadIdx.x] > 0) atomicAdd(migrated, send_count[threadIdx.x]);   // bookkeeping (NbodyLetStats.bodies_migrated)
        send_count[threadIdx.x] = 0;   // ready for the next step's classification
    }
}

// ---- the migrant round with message sizes posted before the host knows the counts (see kernels_let.h)
__global__ void k_let_spec(const int* __restrict__ matrix, const int* __restrict__ pred, int G, int* __restrict__ flags) {
    const int t = threadIdx.x;
    if (t < G * G && (t / G != t % G) && matrix[t] > pred[t]) atomicOr(flags, kFlagMigSpill);   // (the same matrices on every rank: all raise it, or none)
}
__global__ __launch_bounds__(256) void k_let_slot_migrants(const Migrant* __restrict__ packed, const int* __restrict__ send_off,
                                                           const int* __restrict__ pred, int G, int me, Migrant* __restrict__ slots) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= send_off[G]) return;
    int dest = 0, slot0 = 0;
    for (int r = 0; r < G; ++r) {
        if (send_off[r] <= j) dest = r;
    }
    for (int r = 0; r < dest; ++r) slot0 += r == me ? 0 : pred[me * G + r];
    const int k = j - send_off[dest];
    if (dest != me && k < pred[me * G + dest]) slots[slot0 + k] = packed[j];
}
__global__ __launch_bounds__(256) void k_let_append_slots(float4* __restrict__ pos, float4* __restrict__ vel, float4* __restrict__ acc,
                                                          int* __restrict__ ids, int* __restrict__ count, int cap,
                                                          const Migrant* __restrict__ slots_in, const int* __restrict__ matrix,
                                                          const int* __restrict__ pred, int G, int me, int* __restrict__ flags,
                                                          int* __restrict__ new_count) {
    if (*flags & kFlagMigSpill) return;
    const int j = blockIdx.x * 256 + threadIdx.x;   // position in the receive slots
    const int n0 = *count;
    int src = -1, slot0 = 0, before = 0, total = 0;
    for (int r = 0, run = 0; r < G; ++r) {
        const int p = r == me ? 0 : pred[r * G + me], a = r == me ? 0 : matrix[r * G + me];
        if (src < 0 && j < run + p) { src = r; slot0 = run; before = total; }
        run += p;
        total += a;
    }
    if (j == 0) {
        if (n0 + total > cap) atomicOr(flags, kFlagCapacity);
        *new_count = min(cap, n0 + total);
    }
    if (src < 0) return;
    const int k = j - slot0;
    if (k >= matrix[src * G + me]) return;          // the padding of the message
    const int d = n0 + before + k;
    if (d >= cap) return;
    const Migrant m = slots_in[j];
    pos[d] = m.pos; vel[d] = m.vel; acc[d] = m.acc; ids[d] = m.id;
}
__global__ void k_let_report(const int* __restrict__ let_matrix, const int* __restrict__ mig_matrix, const int* __restrict__ offsets,
                             const int* __restrict__ tree_info, const int* __restrict__ flags, int G, int* __restrict__ report) {
    const int gg = G * G;
    for (int t = threadIdx.x; t < gg; t += blockDim.x) { report[t] = let_matrix[t]; report[gg + t] = mig_matrix[t]; }
    for (int t = threadIdx.x; t <= G; t += blockDim.x) report[2 * gg + t] = offsets[t];
    if (threadIdx.x < 3) report[2 * gg + G + 1 + threadIdx.x] = tree_info[threadIdx.x];
    if (threadIdx.x < 4) report[2 * gg + G + 4 + threadIdx.x] = flags[threadIdx.x];
}

// ---- what the other ranks need to know about this one before the second half of the build
// box_ord[0..5] = the overall box, box_ord[6 + 6 b ..] = the box of child cell b, as order-preserving ints (min / max atomics)
__global__ __launch_bounds__(256) void k_let_box(const float4* __restrict__ pos, const float4* __restrict__ acc,
                                                 const unsigned long long* __restrict__ keys, const int* __restrict__ ids,
                                                 const int* __restrict__ count, int* __restrict__ box_ord,
                                                 unsigned long long* __restrict__ weight_sum) {
    __shared__ int red[kBoxes + 1][6];
    __shared__ int wsum_s;
    const int n = *count;
    if (threadIdx.x == 0) wsum_s = 0;
    for (int t = threadIdx.x; t < (kBoxes + 1) * 6; t += 256) red[t / 6][t % 6] = (t % 6) < 3 ? 0x7fffffff : int(0x80000000);
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;   // sorted position
    const bool live = j < n;
    int g = -1, w = 0;
    int o[3] = {0, 0, 0};
    if (live) {
        const int c = min(common_levels(keys[0], keys[n - 1]), kLevels - kBoxDigits);   // the deepest cell holding all my bodies
        g = int((keys[j] >> (3 * (kLevels - kBoxDigits - c))) & (unsigned long long)(kBoxes - 1));
        const float4 p = pos[ids[j]];
        w = body_weight(acc[ids[j]].w);   // what the body cost in the last walk (1 before the first)
        o[0] = f2ord(p.x); o[1] = f2ord(p.y); o[2] = f2ord(p.z);
    }
    // sorted order: the 64 bodies of a wave nearly always share their child cell -- then one lane speaks for all
    const int g0 = __builtin_amdgcn_readfirstlane(g);
    if (__all(live && g == g0)) {
        int lo[3] = {o[0], o[1], o[2]}, hi[3] = {o[0], o[1], o[2]};
        for (int off = 32; off > 0; off >>= 1) {
            for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], __shfl_xor(lo[a], off)); hi[a] = max(hi[a], __shfl_xor(hi[a], off)); }
            w += __shfl_xor(w, off);
        }
        if ((threadIdx.x & 63) == 0) {
            for (int a = 0; a < 3; ++a) {
                atomicMin(&red[g0][a], lo[a]); atomicMax(&red[g0][3 + a], hi[a]);
                atomicMin(&red[kBoxes][a], lo[a]); atomicMax(&red[kBoxes][3 + a], hi[a]);
            }
            atomicAdd(&wsum_s, w);
        }
    } else if (live) {
        for (int a = 0; a < 3; ++a) {
            atomicMin(&red[g][a], o[a]); atomicMax(&red[g][3 + a], o[a]);
            atomicMin(&red[kBoxes][a], o[a]); atomicMax(&red[kBoxes][3 + a], o[a]);
        }
        atomicAdd(&wsum_s, w);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < (kBoxes + 1) * 6; t += 256) {
        const int b = t / 6, a = t % 6, v = red[b][a];
        int* dst = b == kBoxes ? &box_ord[a] : &box_ord[6 + 6 * b + a];
        if (a < 3) { if (v != 0x7fffffff) atomicMin(dst, v); }
        else if (v != int(0x80000000)) atomicMax(dst, v);
    }
    if (threadIdx.x == 0 && wsum_s > 0) atomicAdd(weight_sum, (unsigned long long)wsum_s);
}
__global__ void k_let_ends(const unsigned long long* __restrict__ sorted_keys, const int* __restrict__ count, int* __restrict__ box_ord,
                           unsigned long long* __restrict__ weight_sum, EndInfo* __restrict__ mine) {
    const int n = *count;
    if (threadIdx.x == 0) {
        mine->weight = n > 0 ? (long long)*weight_sum : 0ll;
        *weight_sum = 0ull;   // for the next step
        mine->first_key = n > 0 ? sorted_keys[0] : 0ull;
        mine->last_key = n > 0 ? sorted_keys[n - 1] : 0ull;
        mine->n_bodies = n; mine->pad = 0;
        for (int c = 0; c < 3; ++c) { mine->lo[c] = ord2f(box_ord[c]); mine->hi[c] = ord2f(box_ord[3 + c]); }
    }
    for (int b = threadIdx.x; b < kBoxes; b += blockDim.x)
        for (int c = 0; c < 3; ++c) { mine->box_lo[b][c] = ord2f(box_ord[6 + 6 * b + c]); mine->box_hi[b][c] = ord2f(box_ord[6 + 6 * b + 3 + c]); }
    __syncthreads();
    for (int t = threadIdx.x; t < (kBoxes + 1) * 6; t += blockDim.x) box_ord[t] = (t % 6) < 3 ? 0x7fffffff : int(0x80000000);   // for the next step
}

// edge[0] / edge[1]: levels my first / last sorted body shares with the last / first body of the nearest rank before /
// after me that has bodies (-1: there is none); edge[2] = levels my LAST body shares with that previous rank's last body
// (the cells on my last body's path down to that depth are opened by an earlier rank)
__global__ void k_let_edges(const EndInfo* __restrict__ ends, int G, int me, int* __restrict__ edge) {
    if (threadIdx.x != 0) return;
    int prev = -1, next = -1;
    for (int r = me - 1; r >= 0; --r) if (ends[r].n_bodies > 0) { prev = r; break; }
    for (int r = me + 1; r < G; ++r) if (ends[r].n_bodies > 0) { next = r; break; }
    const bool have = ends[me].n_bodies > 0;
    edge[0] = (have && prev >= 0) ? common_levels(ends[prev].last_key, ends[me].first_key) : -1;
    edge[1] = (have && next >= 0) ? common_levels(ends[me].last_key, ends[next].first_key) : -1;
    edge[2] = (have && prev >= 0) ? common_levels(ends[prev].last_key, ends[me].last_key) : -1;
}

// ---- after the local emit: my RoundB record (node count, flags, what I add to every spanning cell)
__global__ __launch_bounds__(64) void k_let_contrib(const unsigned long long* __restrict__ keys, const signed char* __restrict__ delta,
                                                    const int* __restrict__ base, const Sum4* __restrict__ incl,
                                                    const int* __restrict__ count, const int* __restrict__ info /* {n_nodes, flags} */,
                                                    const EndInfo* __restrict__ ends, const int* __restrict__ edge, int G, int me,
                                                    RoundB* __restrict__ mine, const int* __restrict__ wpre, int by_work,
                                                    const int* __restrict__ own_flags) {
    const int r = blockIdx.x;      // the rank whose spanning cells are concerned
    const int d = threadIdx.x;     // depth
    const int n = *count;
    // (what this rank has raised so far travels too: every rank ends the step with the same verdict, none is left waiting
    // in the next collective for one that has returned an error)
    if (r == 0 && d == 0) { mine->n_nodes = n > 0 ? info[0] : 0; mine->flags = info[1] | *own_flags; mine->pad[0] = mine->pad[1] = 0; }
    if (r == 0 && d < kMaxRanks) {
        // Next step's bounds.  The world's sorted order is the concatenation of the ranks' sorted orders, so its G-quantiles
        // are plain look-ups: quantile j sits at global position j N / G, on the rank whose run of positions contains it.
        // By weight (the walk's visit counts of the last step): the same with the running sum of the weights in the place
        // of the position -- the ranks then get equal WORK, not equal body counts.
        long long before = 0, N = 0, mine_n = n;
        for (int q = 0; q < G; ++q) {
            const long long c = by_work ? ends[q].weight : (long long)ends[q].n_bodies;
            if (q < me) before += c;
            if (q == me) mine_n = c;
            N += c;
        }
        unsigned long long b = 0ull;
        if (d >= 1 && d < G && N > 0 && n > 0) {
            const long long t = min(N - 1, (long long)d * N / G);
            if (t >= before && t < before + mine_n) {
                int k = int(t - before);
                if (by_work) {          // the body whose run of the weight line contains t: last k with wpre[k] <= t - before
                    const int tw = int(t - before);
                    int a = 0, e = n - 1;
                    while (a < e) { const int mid = (a + e + 1) >> 1; if (wpre[mid] <= tw) a = mid; else e = mid - 1; }
                    k = a;
                }
                b = keys[min(k, n - 1)];
            }
        }
        mine->new_bound[d] = b;
    }
    if (d >= kLevels) return;
    Contrib c;
    c.m = c.mx = c.my = c.mz = 0.0; c.cnt = 0; c.base_after = -1;
    if (n > 0 && r < me && ends[r].n_bodies > 0) {
        // my bodies inside the cell of depth d on rank r's last body's path: those with key <= the cell's upper key
        const unsigned long long hi = prefix_hi(ends[r].last_key, d);
        int a = 0, b = n;          // t = number of my keys <= hi
        while (a < b) { const int mid = (a + b) >> 1; if (keys[mid] <= hi) a = mid + 1; else b = mid; }
        const int t = a;
        c.cnt = t;
        if (t > 0) { const Sum4 s = incl[t - 1]; c.m = s.m; c.mx = s.x; c.my = s.y; c.mz = s.z; }
        c.base_after = t < n ? base[t] : info[0];
    } else if (n > 0 && r == me) {
        // is the cell of depth d on MY last body's path one of mine?  (opened by one of my bodies: not shared with the
        // previous rank's last body; spanning: shared with the next rank's first body)
        if (d > edge[2] && d <= edge[1]) {
            const unsigned long long lo = prefix_lo(keys[n - 1], d);
            int a = 0, b = n - 1;  // first of my bodies in the cell
            while (a < b) { const int mid = (a + b) >> 1; if (keys[mid] >= lo) b = mid; else a = mid + 1; }
            const int ko = a;
            const Sum4 up = incl[n - 1];
            const Sum4 bf = ko > 0 ? incl[ko - 1] : Sum4{0.0, 0.0, 0.0, 0.0};
            c.cnt = n - ko;
            c.m = up.m - bf.m; c.mx = up.x - bf.x; c.my = up.y - bf.y; c.mz = up.z - bf.z;
            const int dprev = ko > 0 ? delta[ko - 1] : edge[0];
            c.base_after = base[ko] + (d - (dprev + 1));   // its index in my slice
        }
    }
    mine->c[r][d] = c;
}

// ---- after the all-gather of the RoundB records
// offsets[r] = first global index of rank r's slice, offsets[G] = total
__global__ void k_let_offsets(const RoundB* __restrict__ rb, int G, int* __restrict__ offsets, int* __restrict__ out_flags,
                              unsigned long long* __restrict__ bounds) {
    if (threadIdx.x != 0) return;
    long long run = 0;
    int fl = 0;
    for (int q = 0; q < G; ++q) { offsets[q] = int(run); run += rb[q].n_nodes; fl |= rb[q].flags; }
    offsets[G] = int(run);
    if (run > 0x7fffffffLL) fl |= kFlagNodeCap;   // (global node indices are 32-bit)
    if (fl) atomicOr(out_flags, fl);
    // the bounds the NEXT classification uses (this step's is done): whoever held quantile j reported its key
    unsigned long long prev = 0ull;
    bounds[0] = 0ull;
    for (int j = 1; j < G; ++j) {
        unsigned long long b = 0ull;
        for (int q = 0; q < G; ++q) b = max(b, rb[q].new_bound[j]);
        prev = max(prev, b);     // (monotone; an empty world leaves them all 0)
        bounds[j] = prev;
    }
    bounds[G] = 1ull << 63;
}
// top[r][d] = the finished spanning cells: their records (GLOBAL skip links) go to a small table every rank holds,
// top_nodes[r * kLevels + d]; my own ones also get their finished centre of mass in my slice (after its emit), which the
// opening masks below read.  top_index[r][d] = the cell's global index, -1: none.
__global__ __launch_bounds__(64) void k_let_finalize(const RoundB* __restrict__ rb, const EndInfo* __restrict__ ends, int G, int me,
                                                     float width, float4* __restrict__ slice, const int* __restrict__ offsets,
                                                     int* __restrict__ top_index /* [G][kLevels] */, float4* __restrict__ top_nodes /* [G][kLevels][2] */) {
    const int r = blockIdx.x, d = threadIdx.x;
    if (d >= kLevels) return;
    const int total = offsets[G];
    const Contrib own = rb[r].c[r][d];
    int gi = -1;
    if (own.base_after >= 0) {
        double m = own.m, mx = own.mx, my = own.my, mz = own.mz;
        int skip = total;
        for (int q = r + 1; q < G; ++q) {     // the later ranks, in order, until one still has a body beyond the cell
            if (ends[q].n_bodies <= 0) continue;
            const Contrib c = rb[q].c[r][d];
            m += c.m; mx += c.mx; my += c.my; mz += c.mz;
            if (c.cnt < ends[q].n_bodies) { skip = offsets[q] + c.base_after; break; }
        }
        float w = width;
        for (int q = 0; q < d; ++q) w = w * 0.5f;    // create_orthant halves the width exactly
        gi = offsets[r] + own.base_after;
        const float4 A = make_float4(float(mx / m), float(my / m), float(mz / m), float(m));
        top_nodes[2 * (r * kLevels + d)] = A;
        top_nodes[2 * (r * kLevels + d) + 1] = make_float4(w * w, __int_as_float(skip), __int_as_float(gi), __int_as_float(-1));
        if (r == me) slice[2 * own.base_after] = A;
    }
    top_index[r * kLevels + d] = gi;
}

// ---- which of my private nodes could a partner's bodies reach?  (parent[] / depth[] of the slice come from the emit)
// upper_ok[d] = bit mask of the partners that can open EVERY spanning cell on my first body's path from the root down
// to depth d (those cells belong to earlier ranks); upper_ok[-1] := all.  One thread per (partner, child-cell box): its
// box sits in registers while the block goes down the <= 21 cells.
__global__ __launch_bounds__(kMaxRanks * kBoxes) void k_let_upper(const float4* __restrict__ top_nodes, const int* __restrict__ top_index,
                                                                  const EndInfo* __restrict__ ends, const int* __restrict__ edge, int G, int me,
                                                                  float theta2, unsigned int* __restrict__ upper_ok) {
    __shared__ unsigned int cell_ok[kLevels];   // partners that could open the spanning cell of depth d above my first body
    __shared__ int cell_ti[kLevels];            // its place in top_nodes, -1: none
    const int r = threadIdx.x / kBoxes, b = threadIdx.x % kBoxes;
    const int e0 = edge[0];
    if (threadIdx.x < kLevels) {
        const int d = threadIdx.x;
        cell_ok[d] = 0u;
        int ti = -1;
        if (d <= e0) {   // the cell of depth d that contains my first body: one of the spanning cells of an earlier rank
            const unsigned long long fk = ends[me].first_key;
            for (int q = 0; q < me && ti < 0; ++q)
                if (top_index[q * kLevels + d] >= 0 && ends[q].n_bodies > 0 && prefix_lo(ends[q].last_key, d) == prefix_lo(fk, d))
                    ti = q * kLevels + d;
        }
        cell_ti[d] = ti;
    }
    __syncthreads();
    const bool partner = r < G && r != me && ends[r].n_bodies > 0;
    float lo[3] = {1.f, 1.f, 1.f}, hi[3] = {0.f, 0.f, 0.f};
    if (partner) for (int c = 0; c < 3; ++c) { lo[c] = ends[r].box_lo[b][c]; hi[c] = ends[r].box_hi[b][c]; }
    if (partner && lo[0] <= hi[0])
        for (int d = 0; d <= e0 && d < kLevels; ++d) {
            const int ti = cell_ti[d];
            if (ti >= 0 && box_could_open(top_nodes[2 * ti], top_nodes[2 * ti + 1].x, lo, hi, theta2)) atomicOr(&cell_ok[d], 1u << r);
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int mask = 0;
        for (int q = 0; q < G; ++q) if (q != me && ends[q].n_bodies > 0) mask |= 1u << q;
        for (int d = 0; d < kLevels; ++d) { if (cell_ti[d] >= 0) mask &= cell_ok[d]; upper_ok[d] = mask; }
    }
}

// link[i] = {parent of my node i, the partners that could open it (0 for a leaf)}: one pass over the slice, so that the
// ancestor walk below only ANDs words (testing every ancestor of every node against the boxes was 2 ms at 776 000 nodes).
// The partners' non-empty boxes are staged in LDS once per block.
__global__ __launch_bounds__(256) void k_let_open_masks(const float4* __restrict__ slice, const int* __restrict__ info,
                                                        const EndInfo* __restrict__ ends, int G, int me,
                                                        float theta2, const int* __restrict__ parent, int2* __restrict__ link) {
    __shared__ float outer[kMaxRanks][6];
    __shared__ float sub[kMaxRanks][kBoxes][6];
    __shared__ int n_sub[kMaxRanks];
    if (int(blockIdx.x) * 256 >= info[0]) return;
    if (threadIdx.x < kMaxRanks) n_sub[threadIdx.x] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < G * kBoxes; t += 256) {
        const int r = t / kBoxes, b = t % kBoxes;
        if (r == me || ends[r].n_bodies <= 0) continue;
        if (b == 0) for (int c = 0; c < 3; ++c) { outer[r][c] = ends[r].lo[c]; outer[r][3 + c] = ends[r].hi[c]; }
        if (ends[r].box_lo[b][0] <= ends[r].box_hi[b][0]) {
            const int slot = atomicAdd(&n_sub[r], 1);
            for (int c = 0; c < 3; ++c) { sub[r][slot][c] = ends[r].box_lo[b][c]; sub[r][slot][3 + c] = ends[r].box_hi[b][c]; }
        }
    }
    __syncthreads();
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= info[0]) return;
    const float4 B = slice[2 * idx + 1];
    unsigned int mask = 0;
    if (__float_as_int(B.w) < 0) {      // a cell
        const float4 A = slice[2 * idx];
        for (int r = 0; r < G; ++r) {
            if (r == me || n_sub[r] == 0) continue;
            if (!box_could_open(A, B.x, &outer[r][0], &outer[r][3], theta2)) continue;
            bool open = theta2 == 0.f;  // pruning off
            for (int b = 0; b < n_sub[r] && !open; ++b) open = box_could_open(A, B.x, &sub[r][b][0], &sub[r][b][3], theta2);
            if (open) mask |= 1u << r;
        }
    }
    link[idx] = make_int2(parent[idx], int(mask));
}

// A node goes to partner r if r could open every one of its ancestors; my own spanning cells go to everybody.  The lists
// are written IN NODE ORDER (the receiver finds a node of a list by bisection on its global index): pass 1 works out every
// node's partners and counts them per block of 1 024 nodes, one workgroup turns the counts into every block's first slot
// in every list, pass 2 writes the records.
constexpr int kPackThreads = 1024;
__global__ __launch_bounds__(kPackThreads) void k_let_flag_count(const int* __restrict__ offsets, const int* __restrict__ info,
                                                                 const unsigned char* __restrict__ depth, const int* __restrict__ top_index,
                                                                 const EndInfo* __restrict__ ends, const unsigned int* __restrict__ upper_ok,
                                                                 const int2* __restrict__ link, int G, int me,
                                                                 unsigned int* __restrict__ node_mask, int* __restrict__ block_n /* [blocks][kMaxRanks] */) {
    __shared__ int wave_n[kPackThreads / 64][kMaxRanks];
    if (int(blockIdx.x) * kPackThreads >= info[0]) return;
    const int idx = blockIdx.x * kPackThreads + threadIdx.x;
    const int off = offsets[me];
    unsigned int mask = 0;
    if (idx < info[0]) {
        for (int r = 0; r < G; ++r) if (r != me && ends[r].n_bodies > 0) mask |= 1u << r;
        const int d0 = depth[idx];
        if (!(d0 < kLevels && top_index[me * kLevels + d0] == off + idx)) {   // (a spanning cell of mine: everybody gets it, finished)
            int p = link[idx].x;
            while (mask != 0u && p != -1) {
                if (p <= -2) { mask &= upper_ok[-p - 2]; break; }
                const int2 l = link[p];
                mask &= (unsigned int)l.y;
                p = l.x;
            }
        }
        node_mask[idx] = mask;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = 0; r < G; ++r) {
        const unsigned long long vote = __ballot((mask >> r) & 1u);
        if (lane == 0) wave_n[wave][r] = __popcll(vote);
    }
    __syncthreads();
    if (int(threadIdx.x) < G) {
        int total = 0;
        for (int w = 0; w < kPackThreads / 64; ++w) total += wave_n[w][threadIdx.x];
        block_n[blockIdx.x * kMaxRanks + threadIdx.x] = total;
    }
}
// block_n[b][r] -> the first slot of block b in list r (exclusive scan over the blocks, one partner after the other);
// let_count[r] = the list's length
// list_first[r] = where list r starts in the send buffer (the lists lie one after the other, each exactly as long as it is).
// One wave per partner (kMaxRanks waves): its lanes take consecutive runs of blocks, a wave scan joins them -- no barrier
// until the lists' first slots are added up at the end.
__global__ __launch_bounds__(64 * kMaxRanks) void k_let_pack_scan(const int* __restrict__ info, int G, int* __restrict__ block_n, int* __restrict__ let_count,
                                                                  int* __restrict__ list_first /* [G + 1] */) {
    __shared__ int total[kMaxRanks];
    const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_blocks = (info[0] + kPackThreads - 1) / kPackThreads;
    const int per = (n_blocks + 63) / 64;              // consecutive blocks per lane
    const int b0 = lane * per, b1 = min(n_blocks, b0 + per);
    int sum = 0;
    if (r < G) for (int b = b0; b < b1; ++b) sum += block_n[b * kMaxRanks + r];
    int incl = sum;                                     // inclusive scan over the wave's lanes
#pragma unroll
    for (int step = 1; step < 64; step <<= 1) {
        const int v = __shfl_up(incl, step);
        if (lane >= step) incl += v;
    }
    if (r < G) {
        int run = incl - sum;
        for (int b = b0; b < b1; ++b) { const int c = block_n[b * kMaxRanks + r]; block_n[b * kMaxRanks + r] = run; run += c; }
        if (lane == 63) { let_count[r] = incl; total[r] = incl; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int q = 0; q < G; ++q) { list_first[q] = int(min(run, 0x7fffffffLL)); run += total[q]; }
        list_first[G] = int(min(run, 0x7fffffffLL));
    }
}
__global__ __launch_bounds__(kPackThreads) void k_let_pack(const float4* __restrict__ slice, const int* __restrict__ offsets,
                                                           const int* __restrict__ info, const unsigned char* __restrict__ depth,
                                                           const int* __restrict__ top_index, const float4* __restrict__ top_nodes,
                                                           const unsigned int* __restrict__ node_mask, const int* __restrict__ block_first,
                                                           const int* __restrict__ list_first, int G, int me, LetRecord* __restrict__ send, size_t send_cap) {
    __shared__ int wave_n[kPackThreads / 64][kMaxRanks];
    if (int(blockIdx.x) * kPackThreads >= info[0]) return;
    const int idx = blockIdx.x * kPackThreads + threadIdx.x;
    const unsigned int mask = idx < info[0] ? node_mask[idx] : 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = 0; r < G; ++r) {
        const unsigned long long vote = __ballot((mask >> r) & 1u);
        if (lane == 0) wave_n[wave][r] = __popcll(vote);
    }
    __syncthreads();
    if (int(threadIdx.x) < G) {          // thread r: every wave's first slot in list r
        const int r = threadIdx.x;
        int first = list_first[r] + block_first[blockIdx.x * kMaxRanks + r];
        for (int w = 0; w < kPackThreads / 64; ++w) { const int c = wave_n[w][r]; wave_n[w][r] = first; first += c; }
    }
    __syncthreads();
    if (mask == 0u) return;
    const int off = offsets[me];
    LetRecord rec;
    rec.a = slice[2 * idx]; rec.b = slice[2 * idx + 1];
    const int d0 = depth[idx];
    int skip = off + __float_as_int(rec.b.y);                      // the slice's links are local; the wire carries global ones
    if (d0 < kLevels && top_index[me * kLevels + d0] == off + idx) skip = __float_as_int(top_nodes[2 * (me * kLevels + d0) + 1].y);
    rec.b.y = __int_as_float(skip);
    rec.b.z = __int_as_float(off + idx);   // the slot of NodeB::hot carries the global index (the plain walk does not read it)
    for (int r = 0; r < G; ++r) {
        const bool mine = (mask >> r) & 1u;
        const unsigned long long vote = __ballot(mine);
        if (mine) {
            const size_t slot = size_t(wave_n[wave][r]) + size_t(__popcll(vote & ((1ull << lane) - 1ull)));
            if (slot < send_cap) send[slot] = rec;   // (a buffer that is too small: the host sees it in the counts, grows it and packs again)
        }
    }
}

// ---- the array the walk runs over: the nodes this rank HOLDS -- its slice and what the partners sent -- in the order of
// their global indices, addressed by their position in that order.  in_n[q] = records received from rank q (in the staging
// buffer one rank after the other, each list in node order); layout = {in_at[G], base[G], own_base, n_own, total}.
// A link to global index t becomes the number of held nodes with a smaller index: exact when t is held -- and a node a
// body can get to IS held (it is a child of a cell the body opened) -- and the next held node otherwise (a link nobody
// follows).  A leaf's link is "the next node", which that rule keeps (the DIRECT walk tells leaves by it).
struct HaloLayout { int in_at[kMaxRanks], base[kMaxRanks], own_base, n_own, total, pad; };
__global__ void k_let_layout(const int* __restrict__ in_n, const int* __restrict__ info, int G, int me, HaloLayout* __restrict__ lay, int* __restrict__ split) {
    if (threadIdx.x != 0) return;
    int at = 0, pos = 0;
    const int n_own = info[0];
    for (int q = 0; q < G; ++q) {
        if (q == me) { lay->own_base = pos; pos += n_own; lay->in_at[q] = at; lay->base[q] = lay->own_base; continue; }
        lay->in_at[q] = at; lay->base[q] = pos;
        at += in_n[q]; pos += in_n[q];
    }
    lay->n_own = n_own; lay->total = pos; lay->pad = 0;
    split[0] = 0; split[1] = pos; split[2] = 0; split[3] = 0;
}
__device__ __forceinline__ int held_before(int t, const int* __restrict__ offsets, const int* __restrict__ in_n, const HaloLayout& lay,
                                           const LetRecord* __restrict__ staged, int G, int me) {
    int o = 0;
    while (o < G && offsets[o + 1] <= t) ++o;       // the rank whose slice holds index t (G: t is the end of the tree)
    if (o >= G) return lay.total;
    if (o == me) return lay.own_base + (t - offsets[me]);
    const LetRecord* list = staged + lay.in_at[o];
    int a = 0, b = in_n[o];                          // first record of the list with global index >= t
    while (a < b) {
        const int mid = (a + b) >> 1;
        if (__float_as_int(list[mid].b.z) >= t) b = mid; else a = mid + 1;
    }
    return lay.base[o] + a;
}
__global__ __launch_bounds__(256) void k_let_place_own(const float4* __restrict__ slice, const HaloLayout* __restrict__ lay_p, float4* __restrict__ held) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int n_own = lay_p->n_own, base = lay_p->own_base;
    if (idx >= n_own) return;
    float4 B = slice[2 * idx + 1];
    B.y = __int_as_float(__float_as_int(B.y) + base);
    held[2 * size_t(base + idx)] = slice[2 * idx];
    held[2 * size_t(base + idx) + 1] = B;
}
__global__ __launch_bounds__(256) void k_let_place_halo(const LetRecord* __restrict__ staged, const int* __restrict__ in_n,
                                                        const int* __restrict__ offsets, const HaloLayout* __restrict__ lay_p, int G, int me,
                                                        float4* __restrict__ held) {
    __shared__ HaloLayout lay;
    if (threadIdx.x == 0) lay = *lay_p;
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= lay.total - lay.n_own) return;
    int q = 0;                                       // the list record j belongs to
    for (int r = 0; r < G; ++r) if (r != me && in_n[r] > 0 && lay.in_at[r] <= j) q = r;
    LetRecord rec = staged[j];
    const int pos = lay.base[q] + (j - lay.in_at[q]);
    rec.b.y = __int_as_float(held_before(__float_as_int(rec.b.y), offsets, in_n, lay, staged, G, me));
    rec.b.z = __int_as_float(0);
    held[2 * size_t(pos)] = rec.a;
    held[2 * size_t(pos) + 1] = rec.b;
}
// the walk's node-range segments start at GLOBAL indices total * k / n_split, wherever those fall in what this rank holds:
// the cuts -- and with them the grouping of every body's partial sums -- do not depend on what was exported beyond need
__global__ void k_let_seg_first(const LetRecord* __restrict__ staged, const int* __restrict__ in_n, const int* __restrict__ offsets,
                                const HaloLayout* __restrict__ lay_p, int G, int me, int n_split, int* __restrict__ first) {
    const int k = threadIdx.x;
    if (k > n_split) return;
    const HaloLayout lay = *lay_p;
    const int t = int((long long)offsets[G] * k / n_split);
    first[k] = k == n_split ? lay.total : held_before(t, offsets, in_n, lay, staged, G, me);
}
// my own spanning cells: their links lead into later ranks' slices (after k_let_place_own, which gave them a local link)
__global__ void k_let_place_top(const float4* __restrict__ top_nodes, const int* __restrict__ top_index, const LetRecord* __restrict__ staged,
                                const int* __restrict__ in_n, const int* __restrict__ offsets, const HaloLayout* __restrict__ lay_p, int G, int me,
                                float4* __restrict__ held) {
    const int d = threadIdx.x;
    if (d >= kLevels) return;
    const int gi = top_index[me * kLevels + d];
    if (gi < 0) return;
    const HaloLayout lay = *lay_p;
    const int pos = lay.own_base + (gi - offsets[me]);
    float4 B = held[2 * size_t(pos) + 1];
    B.y = __int_as_float(held_before(__float_as_int(top_nodes[2 * (me * kLevels + d) + 1].y), offsets, in_n, lay, staged, G, me));
    held[2 * size_t(pos) + 1] = B;
}

inline dim3 grid_for(int n, int bs) { return dim3((std::max(n, 1) + bs - 1) / bs); }

}  // namespace

void launch_classify(hipStream_t s, const Shard& sh, int n_upper, const float center[3], float width, const unsigned long long* bounds,
                     int G, int me, unsigned char* dest_of, Migrant* send, int* send_count, int* send_off, int* cursor, bool after_drift) {
    if (n_upper > 0)
        hipLaunchKernelGGL(k_let_classify, grid_for(n_upper, 256), dim3(256), 0, s, sh.own_pos(), sh.own_count(), center[0], center[1],
                           center[2], width, bounds, G, me, sh.keep, dest_of, sh.escaped, send_count, after_drift ? 1 : 0);
    hipLaunchKernelGGL(k_let_mig_offsets, dim3(1), dim3(64), 0, s, send_count, G, send_off, cursor);
    if (n_upper > 0)
        hipLaunchKernelGGL(k_let_pack_migrants, grid_for(n_upper, 256), dim3(256), 0, s, sh.own_pos(), sh.vel, sh.acc, sh.ids, sh.own_count(),
                           dest_of, send_off, cursor, send);
}
void launch_append(hipStream_t s, const Shard& sh, const Migrant* recv, int n_in, int G, int* flags, int* new_count, int* send_count) {
    hipLaunchKernelGGL(k_let_append, grid_for(n_in, 256), dim3(256), 0, s, sh.own_pos(), sh.vel, sh.acc, sh.ids, sh.own_count(), sh.seg_cap,
                       recv, n_in, flags, new_count);
    hipLaunchKernelGGL(k_let_commit_count, dim3(1), dim3(64), 0, s, sh.own_count(), new_count, send_count, G, flags + 2, (const int*)nullptr);
}
void launch_spec_check(hipStream_t s, const int* matrix, const int* pred, int G, int* flags) {
    hipLaunchKernelGGL(k_let_spec, dim3(1), dim3(256), 0, s, matrix, pred, G, flags);
}
void launch_slot_migrants(hipStream_t s, const Migrant* packed, int n_packed_upper, const int* send_off, const int* pred, int G, int me, Migrant* slots) {
    if (n_packed_upper > 0) hipLaunchKernelGGL(k_let_slot_migrants, grid_for(n_packed_upper, 256), dim3(256), 0, s, packed, send_off, pred, G, me, slots);
}
void launch_append_slots(hipStream_t s, const Shard& sh, const Migrant* slots_in, int slots_in_upper, const int* matrix, const int* pred, int G, int me,
                         int* flags, int* new_count, int* send_count) {
    hipLaunchKernelGGL(k_let_append_slots, grid_for(slots_in_upper, 256), dim3(256), 0, s, sh.own_pos(), sh.vel, sh.acc, sh.ids, sh.own_count(), sh.seg_cap,
                       slots_in, matrix, pred, G, me, flags, new_count);
    hipLaunchKernelGGL(k_let_commit_count, dim3(1), dim3(64), 0, s, sh.own_count(), new_count, send_count, G, flags + 2, (const int*)flags);
}
void launch_report(hipStream_t s, const int* let_matrix, const int* mig_matrix, const int* offsets, const int* tree_info, const int* flags, int G,
                   int* report) {
    hipLaunchKernelGGL(k_let_report, dim3(1), dim3(256), 0, s, let_matrix, mig_matrix, offsets, tree_info, flags, G, report);
}
void launch_ends(hipStream_t s, const Shard& sh, int n_upper, const unsigned long long* sorted_keys, const int* sorted_ids, int* box_ord,
                 unsigned long long* weight_sum, EndInfo* mine) {
    if (n_upper > 0)
        hipLaunchKernelGGL(k_let_box, grid_for(n_upper, 256), dim3(256), 0, s, sh.own_pos(), sh.acc, sorted_keys, sorted_ids, sh.own_count(),
                           box_ord, weight_sum);
    hipLaunchKernelGGL(k_let_ends, dim3(1), dim3(64), 0, s, sorted_keys, sh.own_count(), box_ord, weight_sum, mine);
}
void launch_edges(hipStream_t s, const EndInfo* ends, int G, int me, int* edge) {
    hipLaunchKernelGGL(k_let_edges, dim3(1), dim3(64), 0, s, ends, G, me, edge);
}
void launch_contrib(hipStream_t s, const Shard& sh, const TreeDevWork& w, const int* info, const EndInfo* ends, const int* edge, int G, int me,
                    RoundB* mine, bool balance_by_work, const int* own_flags) {
    hipLaunchKernelGGL(k_let_contrib, dim3(G), dim3(64), 0, s, w.keys, w.delta, w.base, static_cast<const Sum4*>(w.incl), sh.own_count(), info,
                       ends, edge, G, me, mine, w.wpre, balance_by_work ? 1 : 0, own_flags);
}
void launch_offsets(hipStream_t s, const RoundB* rb, int G, int* offsets, int* out_flags, unsigned long long* bounds) {
    hipLaunchKernelGGL(k_let_offsets, dim3(1), dim3(64), 0, s, rb, G, offsets, out_flags, bounds);
}
void launch_finalize(hipStream_t s, const RoundB* rb, const EndInfo* ends, int G, int me, float width, float4* slice, const int* offsets, int* top_index,
                     float4* top_nodes) {
    hipLaunchKernelGGL(k_let_finalize, dim3(G), dim3(64), 0, s, rb, ends, G, me, width, slice, offsets, top_index, top_nodes);
}
void launch_flags_and_pack(hipStream_t s, int local_cap, const int* info, const int* edge, const float4* slice, const float4* top_nodes,
                           const int* offsets, const int* top_index, const EndInfo* ends, int G, int me, float theta2, const int* parent,
                           const unsigned char* depth, unsigned int* upper_ok, int2* link, unsigned int* node_mask, int* block_n, int* let_count,
                           int* list_first, LetRecord* send, size_t send_cap, bool prune) {
    // prune = false: theta2 = 0 makes every node "openable": every private node goes to every partner (the test switch
    // that shows the pruning changes nothing but the volume)
    const float t2 = prune ? theta2 : 0.f;
    hipLaunchKernelGGL(k_let_upper, dim3(1), dim3(kMaxRanks * kBoxes), 0, s, top_nodes, top_index, ends, edge, G, me, t2, upper_ok);
    hipLaunchKernelGGL(k_let_open_masks, grid_for(local_cap, 256), dim3(256), 0, s, slice, info, ends, G, me, t2, parent, link);
    hipLaunchKernelGGL(k_let_flag_count, grid_for(local_cap, kPackThreads), dim3(kPackThreads), 0, s, offsets, info, depth, top_index, ends, upper_ok,
                       link, G, me, node_mask, block_n);
    hipLaunchKernelGGL(k_let_pack_scan, dim3(1), dim3(64 * kMaxRanks), 0, s, info, G, block_n, let_count, list_first);
    launch_pack(s, local_cap, info, slice, top_nodes, offsets, top_index, depth, node_mask, block_n, list_first, G, me, send, send_cap);
}
void launch_pack(hipStream_t s, int local_cap, const int* info, const float4* slice, const float4* top_nodes, const int* offsets, const int* top_index,
                 const unsigned char* depth, const unsigned int* node_mask, const int* block_first, const int* list_first, int G, int me, LetRecord* send,
                 size_t send_cap) {
    hipLaunchKernelGGL(k_let_pack, grid_for(local_cap, kPackThreads), dim3(kPackThreads), 0, s, slice, offsets, info, depth, top_index, top_nodes,
                       node_mask, block_first, list_first, G, me, send, send_cap);
}
size_t pack_blocks(int local_cap) { return size_t(grid_for(local_cap, kPackThreads).x); }
void launch_assemble(hipStream_t s, const float4* slice, int local_cap, const LetRecord* staged, int staged_upper, const int* in_n, const int* info,
                     const int* offsets, const int* top_index, const float4* top_nodes, int G, int me, void* layout, int* split, float4* held,
                     int n_split, int* seg_first) {
    HaloLayout* lay = static_cast<HaloLayout*>(layout);
    hipLaunchKernelGGL(k_let_layout, dim3(1), dim3(64), 0, s, in_n, info, G, me, lay, split);
    hipLaunchKernelGGL(k_let_place_own, grid_for(local_cap, 256), dim3(256), 0, s, slice, lay, held);
    if (staged_upper > 0)
        hipLaunchKernelGGL(k_let_place_halo, grid_for(staged_upper, 256), dim3(256), 0, s, staged, in_n, offsets, lay, G, me, held);
    hipLaunchKernelGGL(k_let_place_top, dim3(1), dim3(64), 0, s, top_nodes, top_index, staged, in_n, offsets, lay, G, me, held);
    if (n_split > 1) hipLaunchKernelGGL(k_let_seg_first, dim3(1), dim3(128), 0, s, staged, in_n, offsets, lay, G, me, n_split, seg_first);
}
size_t layout_bytes() { return sizeof(HaloLayout); }

}  // namespace let
}  // namespace nbody
