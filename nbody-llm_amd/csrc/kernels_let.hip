// kernels_let.hip -- Barnes-Hut over SPATIAL shards with a halo ("locally essential tree") exchange
// (BASELINE.json configs[4]: "spatial shard 8 x MI355X with halo cell exchange").
//
// The tree is the one the single-shard device build makes (kernels_tree.hip: same cells, pre-order, skip links as
// BarnesHutSimulation::build_tree, src/manual/barnes_hut.rs:143-183), but no rank ever sees all bodies:
//   * ownership by Morton-key range: rank r holds the bodies whose 63-bit key (the orthant codes of 21 levels) lies in
//     [bound[r], bound[r+1]); the bounds are the G-quantiles of the keys at upload.  Bodies that drift across a bound
//     migrate (k_let_classify packs them per destination, k_let_append takes them in) -- a handful per step;
//   * then the GLOBAL sorted order is the concatenation of the ranks' local sorted orders, so in the build's
//     formulation (body k opens the cells of depths delta[k-1]+1 .. delta[k]; its leaf sits at max(..)+1) only a rank's
//     FIRST and LAST sorted body have a neighbour elsewhere: two "edge" values (k_let_edges) from an all-gather of every
//     rank's first and last key, and each rank emits a contiguous slice of the global pre-order node array;
//   * a cell is opened by its first body, so a cell spanning ranks belongs to exactly one of them and lies on that
//     rank's LAST body's path: at most 21 per rank.  Its sums need the later ranks' bodies below its upper key and its
//     skip link the first node behind them: every rank works out what it adds to every earlier rank's spanning cells
//     (k_let_contrib), one all-gather of those small tables, and everybody knows all spanning cells (k_let_finalize);
//   * everything else is private to a rank's slice.  A partner needs a private node only if one of its bodies can get
//     there, i.e. if it can OPEN every ancestor: k_let_flags tests the ancestors against the partners' bounding boxes
//     (opening test of barnes_hut.rs:192 with the box's nearest point), k_let_pack writes {global index, record} lists,
//     one variable-size send/recv round, k_let_scatter drops them at their global indices in the receiver's array.
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
__device__ __forceinline__ bool could_open(const float4 A, float w2, const float* lo, const float* hi, float theta2) {
    const float dx = fmaxf(0.f, fmaxf(lo[0] - A.x, A.x - hi[0]));
    const float dy = fmaxf(0.f, fmaxf(lo[1] - A.y, A.y - hi[1]));
    const float dz = fmaxf(0.f, fmaxf(lo[2] - A.z, A.z - hi[2]));
    const float d2 = dx * dx + dy * dy + dz * dz;
    return !(w2 < theta2 * d2 * 0.9999f);
}

// ---- migration: a body whose key left this rank's range goes to the rank that owns it
__global__ __launch_bounds__(256) void k_let_classify(const float4* __restrict__ pos, const float4* __restrict__ vel,
                                                      const float4* __restrict__ acc, const int* __restrict__ ids,
                                                      const int* __restrict__ count, float cx, float cy, float cz, float width,
                                                      const unsigned long long* __restrict__ bounds, int G, int me,
                                                      unsigned char* __restrict__ keep, int* __restrict__ escaped,
                                                      Migrant* __restrict__ send, int* __restrict__ send_count, int mig_cap,
                                                      int* __restrict__ flags) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= *count) return;
    const float4 p = pos[k];
    const unsigned long long key = key_of(p, cx, cy, cz, width);
    int dest = 0;   // last r with bounds[r] <= key
    for (int r = 1; r < G; ++r) if (bounds[r] <= key) dest = r;
    keep[k] = dest == me ? 1 : 0;      // (every flag is written: the ones drift_half left belong to the indices before its retain)
    if (dest == me) return;            // an emigrant leaves this rank: a second retain pass (k_compact) closes the gap
    atomicAdd(escaped, 1);
    const int slot = atomicAdd(&send_count[dest], 1);
    if (slot >= mig_cap) { atomicOr(flags, kFlagMigOverflow); return; }
    Migrant m;
    m.pos = p; m.vel = vel[k]; m.acc = acc[k]; m.id = ids[k]; m.pad[0] = m.pad[1] = m.pad[2] = 0;
    send[size_t(dest) * mig_cap + slot] = m;
}

// received migrants (slot s: from rank s) behind the own bodies
__global__ __launch_bounds__(256) void k_let_append(float4* __restrict__ pos, float4* __restrict__ vel, float4* __restrict__ acc,
                                                    int* __restrict__ ids, int* __restrict__ count, int cap,
                                                    const Migrant* __restrict__ recv, const int* __restrict__ recv_count, int G,
                                                    int mig_cap, int* __restrict__ flags, int* __restrict__ new_count) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    int total = 0, src = -1, within = 0;
    for (int s = 0; s < G; ++s) {
        const int c = min(recv_count[s], mig_cap);
        if (src < 0 && j < total + c) { src = s; within = j - total; }
        total += c;
    }
    const int n0 = *count;
    if (j == 0) {
        if (n0 + total > cap) atomicOr(flags, kFlagCapacity);
        *new_count = min(cap, n0 + total);   // (applied by k_let_commit_count: every thread of this launch reads the old count)
    }
    if (src < 0) return;
    const int d = n0 + j;
    if (d >= cap) return;
    const Migrant m = recv[size_t(src) * mig_cap + within];
    pos[d] = m.pos; vel[d] = m.vel; acc[d] = m.acc; ids[d] = m.id;
}
__global__ void k_let_commit_count(int* __restrict__ count, const int* __restrict__ new_count, int* __restrict__ send_count, int G) {
    if (threadIdx.x == 0) *count = *new_count;
    if (int(threadIdx.x) < G) send_count[threadIdx.x] = 0;   // ready for the next step's classification
}

// ---- what the other ranks need to know about this one before the second half of the build
__global__ __launch_bounds__(256) void k_let_box(const float4* __restrict__ pos, const int* __restrict__ count, int* __restrict__ box_ord) {
    __shared__ int red[6][256];
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int n = *count;
    int v[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, int(0x80000000), int(0x80000000), int(0x80000000)};
    if (k < n) {
        const float4 p = pos[k];
        v[0] = v[3] = f2ord(p.x); v[1] = v[4] = f2ord(p.y); v[2] = v[5] = f2ord(p.z);
    }
    for (int c = 0; c < 6; ++c) red[c][threadIdx.x] = v[c];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (int(threadIdx.x) < off)
            for (int c = 0; c < 6; ++c)
                red[c][threadIdx.x] = c < 3 ? min(red[c][threadIdx.x], red[c][threadIdx.x + off]) : max(red[c][threadIdx.x], red[c][threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x < 3) atomicMin(&box_ord[threadIdx.x], red[threadIdx.x][0]);
    else if (threadIdx.x < 6) atomicMax(&box_ord[threadIdx.x], red[threadIdx.x][0]);
}
__global__ void k_let_ends(const unsigned long long* __restrict__ sorted_keys, const int* __restrict__ count, int* __restrict__ box_ord,
                           EndInfo* __restrict__ mine) {
    if (threadIdx.x != 0) return;
    const int n = *count;
    EndInfo e;
    e.first_key = n > 0 ? sorted_keys[0] : 0ull;
    e.last_key = n > 0 ? sorted_keys[n - 1] : 0ull;
    e.n_bodies = n; e.pad = 0;
    for (int c = 0; c < 3; ++c) { e.lo[c] = ord2f(box_ord[c]); e.hi[c] = ord2f(box_ord[3 + c]); }
    *mine = e;
    for (int c = 0; c < 3; ++c) { box_ord[c] = 0x7fffffff; box_ord[3 + c] = int(0x80000000); }   // for the next step
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
                                                    RoundB* __restrict__ mine) {
    const int r = blockIdx.x;      // the rank whose spanning cells are concerned
    const int d = threadIdx.x;     // depth
    const int n = *count;
    if (r == 0 && d == 0) { mine->n_nodes = n > 0 ? info[0] : 0; mine->flags = info[1]; mine->pad[0] = mine->pad[1] = 0; }
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
// offsets[r] = first global index of rank r's slice, offsets[G] = total; top[r][d] = the finished spanning cells
__global__ __launch_bounds__(64) void k_let_finalize(const RoundB* __restrict__ rb, const EndInfo* __restrict__ ends, int G, int me,
                                                     float width, float4* __restrict__ global_nodes, int global_cap,
                                                     int* __restrict__ offsets, int* __restrict__ top_index /* [G][kLevels] global index or -1 */,
                                                     int* __restrict__ out_flags) {
    __shared__ int off_s[kMaxRanks + 1];
    const int r = blockIdx.x, d = threadIdx.x;
    if (d == 0) {
        int run = 0, fl = 0;
        for (int q = 0; q < G; ++q) { off_s[q] = run; run += rb[q].n_nodes; fl |= rb[q].flags; }
        off_s[G] = run;
        if (r == 0) {
            for (int q = 0; q <= G; ++q) offsets[q] = off_s[q];
            if (run > global_cap) fl |= kFlagNodeCap;
            atomicOr(out_flags, fl);
        }
    }
    __syncthreads();
    if (d >= kLevels) return;
    const Contrib own = rb[r].c[r][d];
    int gi = -1;
    if (own.base_after >= 0 && off_s[G] <= global_cap) {
        double m = own.m, mx = own.mx, my = own.my, mz = own.mz;
        int skip = off_s[G];
        for (int q = r + 1; q < G; ++q) {     // the later ranks, in order, until one still has a body beyond the cell
            if (ends[q].n_bodies <= 0) continue;
            const Contrib c = rb[q].c[r][d];
            m += c.m; mx += c.mx; my += c.my; mz += c.mz;
            if (c.cnt < ends[q].n_bodies) { skip = off_s[q] + c.base_after; break; }
        }
        float w = width;
        for (int q = 0; q < d; ++q) w = w * 0.5f;    // create_orthant halves the width exactly
        gi = off_s[r] + own.base_after;
        global_nodes[2 * gi] = make_float4(float(mx / m), float(my / m), float(mz / m), float(m));
        global_nodes[2 * gi + 1] = make_float4(w * w, __int_as_float(skip), __int_as_float(0), __int_as_float(-1));
    }
    top_index[r * kLevels + d] = gi;
}

// my slice -> its place in the global array (indices shifted); must run BEFORE k_let_finalize overwrites the spanning cells
__global__ __launch_bounds__(256) void k_let_place_slice(const float4* __restrict__ local_nodes, const int* __restrict__ info,
                                                         const RoundB* __restrict__ rb, int me, float4* __restrict__ global_nodes,
                                                         int global_cap) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= info[0]) return;
    int off = 0;
    for (int q = 0; q < me; ++q) off += rb[q].n_nodes;
    if (off + i >= global_cap) return;
    const float4 a = local_nodes[2 * i];
    float4 b = local_nodes[2 * i + 1];
    b.y = __int_as_float(__float_as_int(b.y) + off);
    global_nodes[2 * (off + i)] = a;
    global_nodes[2 * (off + i) + 1] = b;
}

// ---- which of my private nodes could a partner's bodies reach?
// parent[i]: index in my slice of node i's parent; -1: none (the root); <= -2: the parent is the spanning cell of depth
// (-p - 2) on my FIRST body's path, which an earlier rank owns
__global__ __launch_bounds__(256) void k_let_parents(const unsigned long long* __restrict__ keys, const signed char* __restrict__ delta,
                                                     const int* __restrict__ base, const int* __restrict__ count,
                                                     const int* __restrict__ info, const int* __restrict__ edge,
                                                     int* __restrict__ parent, unsigned char* __restrict__ depth_out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int n = *count;
    if (n <= 0 || idx >= info[0]) return;
    int lo = 0, hi = n - 1;             // the body whose nodes include idx: last k with base[k] <= idx
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (base[mid] <= idx) lo = mid; else hi = mid - 1; }
    const int k = lo;
    const int d_next = delta[k];
    const int d_prev = k > 0 ? delta[k - 1] : edge[0];
    const int opened = max(0, d_next - d_prev);
    const int t = idx - base[k];
    const int d = t < opened ? d_prev + 1 + t : max(d_prev, d_next) + 1;
    depth_out[idx] = (unsigned char)d;
    int par;
    if (d == 0) par = -1;
    else if (t >= 1) par = idx - 1;     // the cell this body opened one level up
    else {                              // opened by an earlier body: the first one that shares d - 1 levels with k
        const int pd = d - 1;
        const unsigned long long lo_key = prefix_lo(keys[k], pd);
        int a = 0, b = k;
        while (a < b) { const int mid = (a + b) >> 1; if (keys[mid] >= lo_key) b = mid; else a = mid + 1; }
        const int kf = a;
        if (kf == 0 && pd <= edge[0]) par = -(pd + 2);   // ... sits on an earlier rank
        else {
            const int dp = kf > 0 ? delta[kf - 1] : edge[0];
            par = base[kf] + (pd - (dp + 1));
        }
    }
    parent[idx] = par;
}

// upper_ok[d] = bit mask of the partners that can open EVERY spanning cell on my first body's path from the root down
// to depth d (those cells belong to earlier ranks); upper_ok[-1] := all
__global__ void k_let_upper(const float4* __restrict__ global_nodes, const int* __restrict__ top_index, const EndInfo* __restrict__ ends,
                            const int* __restrict__ edge, int G, int me, float theta2, unsigned int* __restrict__ upper_ok) {
    if (threadIdx.x != 0) return;
    unsigned int mask = 0;
    for (int r = 0; r < G; ++r) if (r != me && ends[r].n_bodies > 0) mask |= 1u << r;
    const unsigned long long fk = ends[me].first_key;
    for (int d = 0; d < kLevels; ++d) {
        if (d <= edge[0]) {
            // the cell of depth d that contains my first body: one of the spanning cells of an earlier rank
            int gi = -1;
            for (int r = 0; r < me && gi < 0; ++r)
                if (top_index[r * kLevels + d] >= 0 && ends[r].n_bodies > 0 && prefix_lo(ends[r].last_key, d) == prefix_lo(fk, d))
                    gi = top_index[r * kLevels + d];
            if (gi >= 0) {
                const float4 A = global_nodes[2 * gi], B = global_nodes[2 * gi + 1];
                for (int r = 0; r < G; ++r)
                    if ((mask >> r) & 1u) if (!could_open(A, B.x, ends[r].lo, ends[r].hi, theta2)) mask &= ~(1u << r);
            }
        }
        upper_ok[d] = mask;
    }
}

__global__ __launch_bounds__(256) void k_let_flags(const float4* __restrict__ global_nodes, const int* __restrict__ offsets,
                                                   const int* __restrict__ info, const int* __restrict__ parent,
                                                   const unsigned char* __restrict__ depth, const int* __restrict__ top_index,
                                                   const EndInfo* __restrict__ ends, const unsigned int* __restrict__ upper_ok,
                                                   int G, int me, float theta2, unsigned int* __restrict__ flags,
                                                   int* __restrict__ let_count) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= info[0]) return;
    const int off = offsets[me];
    // my own spanning cells are known to everybody already
    const int d0 = depth[idx];
    unsigned int mask = 0;
    if (!(d0 < kLevels && top_index[me * kLevels + d0] == off + idx)) {
        for (int r = 0; r < G; ++r) if (r != me && ends[r].n_bodies > 0) mask |= 1u << r;
        int p = parent[idx];
        while (mask != 0u && p != -1) {
            if (p <= -2) { mask &= upper_ok[-p - 2]; break; }
            const float4 A = global_nodes[2 * (off + p)], B = global_nodes[2 * (off + p) + 1];
            for (int r = 0; r < G; ++r)
                if ((mask >> r) & 1u) if (!could_open(A, B.x, ends[r].lo, ends[r].hi, theta2)) mask &= ~(1u << r);
            p = parent[p];
        }
    }
    flags[idx] = mask;
    for (int r = 0; r < G; ++r) if ((mask >> r) & 1u) atomicAdd(&let_count[r], 1);
}

__global__ __launch_bounds__(256) void k_let_pack(const float4* __restrict__ global_nodes, const int* __restrict__ offsets,
                                                  const int* __restrict__ info, const unsigned int* __restrict__ flags, int G, int me,
                                                  LetRecord* __restrict__ send, size_t send_stride, int* __restrict__ cursor) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= info[0]) return;
    const unsigned int mask = flags[idx];
    if (mask == 0u) return;
    const int gi = offsets[me] + idx;
    LetRecord rec;
    rec.a = global_nodes[2 * gi]; rec.b = global_nodes[2 * gi + 1]; rec.index = gi; rec.pad[0] = rec.pad[1] = rec.pad[2] = 0;
    for (int r = 0; r < G; ++r)
        if ((mask >> r) & 1u) {
            const int slot = atomicAdd(&cursor[r], 1);   // (the order inside a list does not matter: the receiver scatters by index)
            if (size_t(slot) < send_stride) send[size_t(r) * send_stride + slot] = rec;
        }
}

__global__ __launch_bounds__(256) void k_let_scatter(const LetRecord* __restrict__ recv, int n, float4* __restrict__ global_nodes,
                                                     int global_cap) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const LetRecord rec = recv[j];
    if (rec.index < 0 || rec.index >= global_cap) return;
    global_nodes[2 * rec.index] = rec.a;
    global_nodes[2 * rec.index + 1] = rec.b;
}

inline dim3 grid_for(int n, int bs) { return dim3((std::max(n, 1) + bs - 1) / bs); }

}  // namespace

void launch_classify(hipStream_t s, const Shard& sh, int n_upper, const float center[3], float width, const unsigned long long* bounds,
                     int G, int me, Migrant* send, int* send_count, int mig_cap, int* flags) {
    if (n_upper <= 0) return;
    hipLaunchKernelGGL(k_let_classify, grid_for(n_upper, 256), dim3(256), 0, s, sh.own_pos(), sh.vel, sh.acc, sh.ids, sh.own_count(),
                       center[0], center[1], center[2], width, bounds, G, me, sh.keep, sh.escaped, send, send_count, mig_cap, flags);
}
void launch_append(hipStream_t s, const Shard& sh, const Migrant* recv, const int* recv_count, int G, int mig_cap, int* flags,
                   int* new_count, int* send_count) {
    hipLaunchKernelGGL(k_let_append, grid_for(G * mig_cap, 256), dim3(256), 0, s, sh.own_pos(), sh.vel, sh.acc, sh.ids, sh.own_count(),
                       sh.seg_cap, recv, recv_count, G, mig_cap, flags, new_count);
    hipLaunchKernelGGL(k_let_commit_count, dim3(1), dim3(64), 0, s, sh.own_count(), new_count, send_count, G);
}
void launch_ends(hipStream_t s, const Shard& sh, int n_upper, const unsigned long long* sorted_keys, int* box_ord, EndInfo* mine) {
    if (n_upper > 0) hipLaunchKernelGGL(k_let_box, grid_for(n_upper, 256), dim3(256), 0, s, sh.own_pos(), sh.own_count(), box_ord);
    hipLaunchKernelGGL(k_let_ends, dim3(1), dim3(64), 0, s, sorted_keys, sh.own_count(), box_ord, mine);
}
void launch_edges(hipStream_t s, const EndInfo* ends, int G, int me, int* edge) {
    hipLaunchKernelGGL(k_let_edges, dim3(1), dim3(64), 0, s, ends, G, me, edge);
}
void launch_contrib(hipStream_t s, const Shard& sh, const TreeDevWork& w, const int* info, const EndInfo* ends, const int* edge, int G, int me,
                    RoundB* mine) {
    hipLaunchKernelGGL(k_let_contrib, dim3(G), dim3(64), 0, s, w.keys, w.delta, w.base, static_cast<const Sum4*>(w.incl), sh.own_count(), info,
                       ends, edge, G, me, mine);
}
void launch_finalize(hipStream_t s, const float4* local_nodes, int local_cap, const int* info, const RoundB* rb, const EndInfo* ends, int G,
                     int me, float width, float4* global_nodes, int global_cap, int* offsets, int* top_index, int* out_flags) {
    hipLaunchKernelGGL(k_let_place_slice, grid_for(local_cap, 256), dim3(256), 0, s, local_nodes, info, rb, me, global_nodes, global_cap);
    hipLaunchKernelGGL(k_let_finalize, dim3(G), dim3(64), 0, s, rb, ends, G, me, width, global_nodes, global_cap, offsets, top_index, out_flags);
}
void launch_flags_and_pack(hipStream_t s, const Shard& sh, const TreeDevWork& w, int local_cap, const int* info, const int* edge,
                           const float4* global_nodes, const int* offsets, const int* top_index, const EndInfo* ends, int G, int me,
                           float theta2, int* parent, unsigned char* depth, unsigned int* upper_ok, unsigned int* flags, int* let_count,
                           LetRecord* send, size_t send_stride, int* cursor, bool prune) {
    hipLaunchKernelGGL(k_let_parents, grid_for(local_cap, 256), dim3(256), 0, s, w.keys, w.delta, w.base, sh.own_count(), info, edge, parent, depth);
    // prune = false: theta2 = 0 makes every node "openable": every private node goes to every partner (the test switch
    // that shows the pruning changes nothing but the volume)
    const float t2 = prune ? theta2 : 0.f;
    hipLaunchKernelGGL(k_let_upper, dim3(1), dim3(64), 0, s, global_nodes, top_index, ends, edge, G, me, t2, upper_ok);
    hipLaunchKernelGGL(k_let_flags, grid_for(local_cap, 256), dim3(256), 0, s, global_nodes, offsets, info, parent, depth, top_index, ends,
                       upper_ok, G, me, t2, flags, let_count);
    hipLaunchKernelGGL(k_let_pack, grid_for(local_cap, 256), dim3(256), 0, s, global_nodes, offsets, info, flags, G, me, send, send_stride, cursor);
}
void launch_scatter(hipStream_t s, const LetRecord* recv, int n, float4* global_nodes, int global_cap) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_let_scatter, grid_for(n, 256), dim3(256), 0, s, recv, n, global_nodes, global_cap);
}

}  // namespace let
}  // namespace nbody
