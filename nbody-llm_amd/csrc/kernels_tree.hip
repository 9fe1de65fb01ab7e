// kernels_tree.hip -- device-side octree build, the fast-math alternative to
// octree_host.cpp.  Produces the SAME cells, pre-order and skip links as
// BarnesHutSimulation::build_tree (src/manual/barnes_hut.rs:143-183); only the centre-of-mass sums
// differ in rounding (f64 prefix sums over the sorted bodies instead of the reference's sequential
// f32 folds), which is why the host build stays the default and the strict path.
//
//   1. key[k]  = the body's orthant codes on levels 0..20 (3 bits each), computed with the
//                reference's own recurrences: code bit i set iff p[i] > center[i]
//                (shared.rs:245-254); child half width = hw/2, child centre = centre +- child half
//                width (shared.rs:256-272).  Sorting by key puts the bodies in depth-first leaf order
//                with children in orthant order.
//   2. stable radix sort of (key, id) (rocPRIM).
//   3. delta[k] = common levels of key[k], key[k+1]; body k opens the internal nodes of depths
//                delta[k-1]+1 .. delta[k] (cells that contain k and k+1 but not k-1); its leaf sits at
//                depth max(delta[k-1], delta[k]) + 1 -- the reference splits until a body is alone.
//   4. an exclusive scan of (opened + 1) gives every node its pre-order index; an inclusive scan of
//                {m, m x, m y, m z} in f64 gives every cell's mass and centre of mass by subtraction.
//   5. emit: one thread per body writes its opened cells and its leaf.  A cell's last body is found
//                by binary search on the sorted keys (all keys sharing its prefix), its skip link is
//                the pre-order index after that body's leaf.
// Bodies whose 63-bit keys collide (at N = 2^22 in a Plummer sphere there is about one such pair at any time: cells of
// level 21 are 3e-5 wide) get a SECOND key, levels 21..41, computed and ordered inside their group of equal first keys
// by the group's first thread (k_tree_ties); delta then runs to 42 and the cells below level 21 are found by walking the
// group instead of a binary search.  Bodies that agree on all 42 levels (coincident for f32 purposes: the cell width is
// far below an ulp of the coordinates by then) make the build report "too deep"; the one-GPU caller falls back to the
// host build (which goes to depth 192 before it gives up), the spatial-shard caller returns NBODY_ERR_TREE_DEPTH.
#include "kernels.h"
#include "kernels_f64.h"   // Node64: the F = f64 instantiation of the build writes 64-byte records

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/block/block_radix_sort.hpp>

// test hook: the largest group of equal 63-bit keys the device build orders itself (1: any collision is "too deep", which
// exercises the callers' fallback paths the way every collision did before the second keys existed)

namespace nbody {

namespace {

constexpr int kLevels = 21;

struct Sum4 { double m, x, y, z; };
struct Sum4Plus {
    __host__ __device__ Sum4 operator()(const Sum4& a, const Sum4& b) const { return Sum4{a.m + b.m, a.x + b.x, a.y + b.y, a.z + b.z}; }
};

// the orthant codes of kLevels levels starting at level `first` (0: the sort key; kLevels: the tie-break key)
template <class P4, class Real>
__device__ __forceinline__ unsigned long long orthant_key(const P4 p, Real cx, Real cy, Real cz, Real width, int first) {
    Real hw = width * Real(0.5);  // Bounds::new
    unsigned long long key = 0;
#pragma unroll 1
    for (int l = 0; l < first + kLevels; ++l) {
        const bool bx = p.x > cx, by = p.y > cy, bz = p.z > cz;  // get_orthant
        key = (key << 3) | (unsigned long long)((bx ? 1 : 0) | (by ? 2 : 0) | (bz ? 4 : 0));   // (the levels before `first` fall off the top)
        hw = hw * Real(0.5);                                       // create_orthant
        cx = bx ? cx + hw : cx - hw;
        cy = by ? cy + hw : cy - hw;
        cz = bz ? cz + hw : cz - hw;
    }
    return key & 0x7fffffffffffffffull;
}

// one node record: {com, mass | w^2, skip, hot, body} as two float4 (F = f32) or one Node64 (F = f64)
__device__ __forceinline__ void put_node(float4* nodes, size_t i, float x, float y, float z, float m, float w2, int skip, int hot, int body) {
    nodes[2 * i] = make_float4(x, y, z, m);
    nodes[2 * i + 1] = make_float4(w2, __int_as_float(skip), __int_as_float(hot), __int_as_float(body));
}
__device__ __forceinline__ void put_node(nbody64::Node64* nodes, size_t i, double x, double y, double z, double m, double w2, int skip, int hot,
                                         int body) {
    nbody64::Node64 r;
    r.x = x; r.y = y; r.z = z; r.m = m; r.w2 = w2; r.skip = skip; r.hot = hot; r.body = body; r.pad = 0; r.pad2 = 0.0;
    nodes[i] = r;
}

template <class P4, class Real>
__global__ __launch_bounds__(256) void k_tree_keys(const P4* __restrict__ pos, const int* __restrict__ count,
                                                   int n_upper, Real cx0, Real cy0, Real cz0, Real width,
                                                   unsigned long long* __restrict__ keys, int* __restrict__ ids,
                                                   int* __restrict__ out_info, int* __restrict__ counters) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k == 0) { out_info[0] = 0; out_info[1] = 0; }  // node count and flags of this build (first kernel of the build)
    if (k >= n_upper) return;
    if (k >= *count) {  // n_upper only bounds the live count: the unused tail sorts to the end (bit 63 set),
        keys[k] = ~0ull;  // where every later kernel ignores it (they all read *count)
        ids[k] = k;
        return;
    }
    keys[k] = orthant_key(pos[k], cx0, cy0, cz0, width, 0);
    ids[k] = k;
}

__device__ __forceinline__ int common_levels(unsigned long long a, unsigned long long b) {
    const unsigned long long x = a ^ b;
    if (x == 0) return kLevels;               // identical on all 21 levels
    return (__clzll((long long)x) - 1) / 3;   // bit 63 is unused
}
// levels two sorted neighbours share, 0 .. 2 kLevels (keys2 is only defined inside groups of equal first keys)
__device__ __forceinline__ int common_levels2(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ keys2,
                                              int a, int b) {
    const int c = common_levels(keys[a], keys[b]);
    return c < kLevels ? c : kLevels + common_levels(keys2[a], keys2[b]);
}

// The radix sort covers the top 48 bits of the keys only (16 levels: 6 passes of 8 bits instead of 8 -- two launches and
// a quarter of the histogram less, 12-16 us of a ~100 us build); what it leaves is finished here.  The first body of every
// group of equal TOP bits (bodies that share 16 levels: a handful here and there) sorts the group by the full key
// (insertion sort; the radix sort is stable, so equal keys keep ascending ids), and inside it every run of equal FULL
// keys gets its second keys (levels 21..41) and is put in their order.  Groups are pairs in practice.  A group of more
// than kMaxLowGroup bodies (a dense clump inside one level-16 cell, 1e-3 of a width-64 box) is not sorted by one thread:
// the workgroup its leader sits in sorts it together (sort_big_group: up to kMaxBigGroup bodies, kBigPerBlock such groups per
// workgroup).  Beyond that -- and for a run of more than tuning().tree_max_tie (64) equal FULL keys -- the build
// raises a flag: 1 = "deeper than the build's 42 levels", 4 = "a clump larger than the build sorts" (the single-GPU step
// then builds on the host; a spatial rank reports it as what it is).
constexpr int kSortLowBits = 15;    // key bits the radix sort leaves to k_tree_ties (levels 16..20)
constexpr int kMaxLowGroup = 256;
constexpr int kBigItems = 16;
constexpr int kMaxBigGroup = 256 * kBigItems;   // 4096 bodies: one workgroup's block_radix_sort

// runs of equal FULL keys inside the sorted group [a0, e): second keys (levels 21..41), put in their order
template <class P4, class Real>
__device__ void finish_equal_key_runs(const P4* __restrict__ pos, Real cx0, Real cy0, Real cz0, Real width, unsigned long long* __restrict__ keys,
                                      unsigned long long* __restrict__ keys2, int* __restrict__ ids, int a0, int e, int* __restrict__ flags, int kMaxTie) {
    for (int a = a0; a + 1 < e;) {
        int z = a + 1;
        while (z < e && keys[z] == keys[a]) ++z;
        if (z - a > 1) {
            if (z - a > kMaxTie) { atomicOr(flags, 1); return; }
            for (int q = a; q < z; ++q) keys2[q] = orthant_key(pos[ids[q]], cx0, cy0, cz0, width, kLevels);
            for (int q = a + 1; q < z; ++q) {
                const unsigned long long k2 = keys2[q];
                const int id = ids[q];
                int r = q - 1;
                while (r >= a && keys2[r] > k2) { keys2[r + 1] = keys2[r]; ids[r + 1] = ids[r]; --r; }
                keys2[r + 1] = k2; ids[r + 1] = id;
            }
        }
        a = z;
    }
}

// A big group (257 .. 4096 bodies sharing 16 levels) is sorted by the WORKGROUP its leader sits in: a stable block radix
// sort of {15 low key bits, place in the group} pairs (16 KB of LDS: the 8-byte keys stay in global memory and are
// gathered into their new places afterwards), then the runs of equal full keys as above.
template <class P4, class Real>
__device__ void sort_big_group(const P4* __restrict__ pos, Real cx0, Real cy0, Real cz0, Real width, unsigned long long* __restrict__ keys,
                               unsigned long long* __restrict__ keys2, int* __restrict__ ids, int* __restrict__ flags, int kMaxTie, int b, int e) {
    using Sort = rocprim::block_radix_sort<unsigned short, 256, kBigItems, unsigned short>;
    __shared__ typename Sort::storage_type storage;
    unsigned short k[kBigItems], v[kBigItems];
#pragma unroll
    for (int i = 0; i < kBigItems; ++i) {
        const int q = b + int(threadIdx.x) * kBigItems + i;
        k[i] = q < e ? (unsigned short)(keys[q] & ((1ull << kSortLowBits) - 1ull)) : (unsigned short)0xffff;   // (the padding sorts behind everything)
        v[i] = q < e ? (unsigned short)(int(threadIdx.x) * kBigItems + i) : (unsigned short)0xffff;
    }
    __syncthreads();
    Sort().sort(k, v, storage, 0, 16);
    unsigned long long kk[kBigItems];
    int ii[kBigItems];
#pragma unroll
    for (int i = 0; i < kBigItems; ++i) {
        const bool live = v[i] != (unsigned short)0xffff;
        kk[i] = live ? keys[b + v[i]] : 0ull;
        ii[i] = live ? ids[b + v[i]] : 0;
    }
    __syncthreads();   // (everybody has read its sources before anybody writes)
#pragma unroll
    for (int i = 0; i < kBigItems; ++i) {
        const int q = b + int(threadIdx.x) * kBigItems + i;
        if (q < e) { keys[q] = kk[i]; ids[q] = ii[i]; }
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) finish_equal_key_runs(pos, cx0, cy0, cz0, width, keys, keys2, ids, b, e, flags, kMaxTie);
    __syncthreads();
}

constexpr int kBigPerBlock = 8;   // big groups one workgroup of k_tree_ties can lead

template <class P4, class Real>
__global__ __launch_bounds__(256) void k_tree_ties(const P4* __restrict__ pos, const int* __restrict__ count, Real cx0, Real cy0,
                                                   Real cz0, Real width, unsigned long long* __restrict__ keys,
                                                   unsigned long long* __restrict__ keys2, int* __restrict__ ids, int* __restrict__ flags,
                                                   int kMaxTie) {
    __shared__ int big[2 * kBigPerBlock];
    __shared__ int n_big;
    if (threadIdx.x == 0) n_big = 0;
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int n = *count;
    bool leader = j + 1 < n;
    unsigned long long top = 0;
    if (leader) {
        top = keys[j] >> kSortLowBits;
        leader = (keys[j + 1] >> kSortLowBits) == top && !(j > 0 && (keys[j - 1] >> kSortLowBits) == top);
    }
    if (leader) {
        int e = j + 2;
        while (e < n && e - j <= kMaxBigGroup && (keys[e] >> kSortLowBits) == top) ++e;
        if (e - j > kMaxLowGroup) {
            if (e - j > kMaxBigGroup) atomicOr(flags, 4);
            else {
                const int slot = atomicAdd(&n_big, 1);
                if (slot >= kBigPerBlock) atomicOr(flags, 4);
                else { big[2 * slot] = j; big[2 * slot + 1] = e; }
            }
        } else {
            for (int q = j + 1; q < e; ++q) {           // by the full key
                const unsigned long long k = keys[q];
                const int id = ids[q];
                int r = q - 1;
                while (r >= j && keys[r] > k) { keys[r + 1] = keys[r]; ids[r + 1] = ids[r]; --r; }
                keys[r + 1] = k; ids[r + 1] = id;
            }
            finish_equal_key_runs(pos, cx0, cy0, cz0, width, keys, keys2, ids, j, e, flags, kMaxTie);
        }
    }
    __syncthreads();
    const int todo = min(n_big, kBigPerBlock);   // (uniform: every thread of the workgroup takes part in the sorts)
    for (int g = 0; g < todo; ++g) sort_big_group(pos, cx0, cy0, cz0, width, keys, keys2, ids, flags, kMaxTie, big[2 * g], big[2 * g + 1]);
}

__device__ __forceinline__ void split_anc_block(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ keys2,
                                                const signed char* __restrict__ delta, const int* __restrict__ base, int n, int n_nodes, int n_split,
                                                int* __restrict__ first, int* __restrict__ n_anc, int* __restrict__ anc, int max_anc,
                                                const int* __restrict__ info, int* __restrict__ poison, int node_cap, const int s, const int a);

// One thread per NODE (not per body: the first body of a big cell opens every level above it, and 15
// cells x a 17-step binary search in one thread was the kernel's whole duration, 30 us).  Node idx
// belongs to the body k with base[k] <= idx < base[k+1] (binary search); its t-th node is the cell of
// depth delta[k-1]+1+t that the body opens, or -- the last one -- the body's leaf.
template <class P4, class Real, class NodeT>
__global__ __launch_bounds__(256) void k_tree_emit(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ keys2,
                                                   const int* __restrict__ ids, const P4* __restrict__ pos,
                                                   const int* __restrict__ count, const signed char* __restrict__ delta,
                                                   const int* __restrict__ base, const Sum4* __restrict__ incl,
                                                   Real width, NodeT* __restrict__ nodes, int node_cap,
                                                   int* __restrict__ order, int* __restrict__ out_info, int want_hot,
                                                   const int* __restrict__ edge, const int* __restrict__ node_offset,
                                                   int* __restrict__ parent_out, unsigned char* __restrict__ depth_out,
                                                   TreeSplitReq split, int emit_blocks) {
    if (int(blockIdx.x) >= emit_blocks) {   // the walk's split points, one workgroup each, in the same launch (they need the scans' results only)
        if (threadIdx.x < 64)
            split_anc_block(keys, keys2, delta, base, *count, node_cap, split.n_split, split.first, split.n_anc, split.anc, split.max_anc, split.info,
                            split.poison, node_cap, int(blockIdx.x) - emit_blocks, threadIdx.x);
        return;
    }
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int n = *count;
    // node_offset (spatial shards pass 0 and shift on the wire): indices and links of the slice shifted by it
    const int off = node_offset ? *node_offset : 0;
    if (n == 0) {  // the reference's empty root (barnes_hut.rs:145)
        if (node_offset) { if (idx == 0) { out_info[0] = 0; out_info[2] = 0; } return; }   // (a rank without bodies adds nothing to the world's tree)
        if (idx == 0) {
            if (node_cap >= 1) {
                put_node(nodes, 0, Real(0), Real(0), Real(0), Real(0), width * width, 1, 0, -1);
            }
            out_info[0] = 1;
            out_info[2] = 0;
            if (parent_out) { parent_out[0] = -1; depth_out[0] = 0; }
        }
        return;
    }
    const int total = out_info[0];      // k_tree_scan: the last sorted body's first node + what it emits
    if (off + total > node_cap) { if (idx == 0) atomicOr(out_info + 1, 2); return; }
    if (idx >= total) return;
    int lo = 0, hi = n - 1;             // the body whose nodes include idx: last k with base[k] <= idx
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (base[mid] <= idx) lo = mid; else hi = mid - 1;
    }
    const int k = lo;
    const int d_next = delta[k];
    const int d_prev = (k > 0) ? delta[k - 1] : (edge ? edge[0] : -1);
    const int opened = max(0, d_next - d_prev);
    const int t = idx - base[k];
    const unsigned long long key = keys[k];
    // NodeB::hot (octree_host.h): bodies in the grandparent cell = sorted bodies sharing the first depth-2 levels
    const int my_depth = (t < opened) ? d_prev + 1 + t : max(d_prev, d_next) + 1;
    if (parent_out) {
        // index in the slice of the node's parent; -1: none (the root); <= -2: the cell of depth (-p - 2) on the slice's
        // FIRST body's path that an earlier rank owns (spatial shards)
        int par;
        if (my_depth == 0) par = -1;
        else if (t >= 1) par = idx - 1; // the cell this body opened one level up
        else {                          // opened by an earlier body: the first one that shares my_depth - 1 levels with k
            const int pd = my_depth - 1;
            int a = 0, b = k;
            if (pd > kLevels) {
                a = k;
                while (a > 0 && keys[a - 1] == key && common_levels(keys2[a - 1], keys2[k]) >= pd - kLevels) --a;
            } else {
                const int sh = 3 * (kLevels - pd);
                const unsigned long long lo_key = sh >= 63 ? 0ull : (key >> sh) << sh;
                while (a < b) { const int mid = (a + b) >> 1; if (keys[mid] >= lo_key) b = mid; else a = mid + 1; }
            }
            const int kf = a;
            if (kf == 0 && edge && pd <= edge[0]) par = -(pd + 2);
            else {
                const int dp = kf > 0 ? delta[kf - 1] : (edge ? edge[0] : -1);
                par = base[kf] + (pd - (dp + 1));
            }
        }
        parent_out[idx] = par;
        depth_out[idx] = (unsigned char)my_depth;
    }
    int hot = n;
    if (want_hot && my_depth - 2 > kLevels) {   // below level 21: the grandparent's bodies are neighbours inside the group of equal keys
        const int lv = my_depth - 2 - kLevels;
        int a = k, b = k;
        while (a > 0 && keys[a - 1] == key && common_levels(keys2[a - 1], keys2[k]) >= lv) --a;
        while (b + 1 < n && keys[b + 1] == key && common_levels(keys2[b + 1], keys2[k]) >= lv) ++b;
        hot = b - a + 1;
    } else if (want_hot && my_depth >= 2) {   // (two more binary searches per node, 5 us at N = 65 536: only for the walk that uses it)
        const int sh = 3 * (kLevels - (my_depth - 2));
        const unsigned long long lo_key = (key >> sh) << sh, hi_gp = lo_key | ((1ull << sh) - 1ull);
        int a = 0, b = k;               // first sorted body with key >= lo_key
        while (a < b) { const int mid = (a + b) >> 1; if (keys[mid] >= lo_key) b = mid; else a = mid + 1; }
        const int first = a;
        a = k; b = n - 1;               // last sorted body with key <= hi_gp
        while (a < b) { const int mid = (a + b + 1) >> 1; if (keys[mid] <= hi_gp) a = mid; else b = mid - 1; }
        hot = a - first + 1;
    }
    if (t < opened) {                   // a cell this body opens, shallowest first
        const int d = d_prev + 1 + t;   // depth of the cell: its bodies share d levels
        int a = k, b = n - 1;
        if (d > kLevels) {              // a cell below level 21: its bodies follow k inside the group of equal keys
            while (a + 1 < n && keys[a + 1] == key && common_levels(keys2[k], keys2[a + 1]) >= d - kLevels) ++a;
        } else {
            const int shift = 3 * (kLevels - d);
            const unsigned long long hi_key = key | ((shift >= 64) ? ~0ull : ((1ull << shift) - 1ull));
            while (a < b) {             // last sorted body with key <= hi_key
                const int mid = (a + b + 1) >> 1;
                if (keys[mid] <= hi_key) a = mid; else b = mid - 1;
            }
        }
        const int j = a;
        const Sum4 before = (k > 0) ? incl[k - 1] : Sum4{0.0, 0.0, 0.0, 0.0};
        const Sum4 upto = incl[j];
        const double m = upto.m - before.m;
        Real w = width;
        for (int q = 0; q < d; ++q) w = w * Real(0.5);  // create_orthant halves the width exactly
        const int skip = off + ((j + 1 < n) ? base[j + 1] : total);
        put_node(nodes, size_t(off + idx), Real((upto.x - before.x) / m), Real((upto.y - before.y) / m), Real((upto.z - before.z) / m), Real(m),
                 w * w, skip, hot, -1);
    } else {                            // the body's leaf
        const int ld = max(d_prev, d_next) + 1;
        Real w = width;
        for (int q = 0; q < ld; ++q) w = w * Real(0.5);
        const int id = ids[k];
        const P4 p = pos[id];
        put_node(nodes, size_t(off + idx), p.x, p.y, p.z, p.w, w * w, off + idx + 1, hot, id);
        order[k] = id;
    }
}

// ancestors of the node-range split points (the walk's WalkSplit lists), root first.  Node t lies in
// the run emitted by body k (base[k] <= t); its ancestors are the cells of depths 0 .. depth(t)-1 on
// that body's path, each opened by the first sorted body that shares the prefix.
__device__ __forceinline__ void split_anc_block(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ keys2,
                                                const signed char* __restrict__ delta, const int* __restrict__ base, int n, int n_nodes, int n_split,
                                                int* __restrict__ first, int* __restrict__ n_anc, int* __restrict__ anc, int max_anc,
                                                const int* __restrict__ info, int* __restrict__ poison, int node_cap, const int s, const int a) {
    if (info && node_cap >= 0 && info[1] == 0 && info[0] > node_cap) {   // (riding in the emit's launch: its own "node array too small" may not be visible yet)
        if (poison && s == 0 && a == 0) atomicOr(poison, 2);
        return;
    }
    if (info) {   // unsynchronised step: the host has not seen this build's result
        if (info[1] != 0) {   // the build needs the host (too deep / node array too small): stop everything that follows
            if (poison && s == 0 && a == 0) atomicOr(poison, info[1]);
            return;
        }
        n_nodes = info[0];
        n = info[2];
        if (n <= 0) {   // the reference's empty root: one segment boundary set
            if (a == 0) { first[s] = (s == 0) ? 0 : n_nodes; if (s == n_split - 1) first[n_split] = n_nodes; n_anc[s] = 0; }
            return;
        }
    }
    const int t = int((long long)n_nodes * s / n_split);
    if (a == 0) {
        first[s] = t;
        if (s == n_split - 1) first[n_split] = n_nodes;
    }
    int lo = 0, hi = n - 1;    // body whose run contains node t: last k with base[k] <= t
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (base[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const int k = lo;
    const int d_prev = (k > 0) ? delta[k - 1] : -1;
    const int depth = d_prev + 1 + (t - base[k]);  // cells opened at k have depths d_prev+1.., the leaf follows
    if (a == 0) n_anc[s] = min(depth, max_anc);
    if (a >= depth || a >= max_anc) return;
    const unsigned long long key = keys[k];
    int l2 = 0, h2 = k;        // first sorted body inside the depth-a cell: it opened it
    if (a > kLevels) {         // below level 21: inside the group of equal keys
        l2 = k;
        while (l2 > 0 && keys[l2 - 1] == key && common_levels(keys2[l2 - 1], keys2[k]) >= a - kLevels) --l2;
    } else {
        const int shift = 3 * (kLevels - a);
        const unsigned long long lo_key = (shift >= 64) ? 0ull : (key >> shift) << shift;
        while (l2 < h2) {
            const int mid = (l2 + h2) >> 1;
            if (keys[mid] >= lo_key) h2 = mid; else l2 = mid + 1;
        }
    }
    const int kf = l2;
    const int dp = (kf > 0) ? delta[kf - 1] : -1;
    anc[s * max_anc + a] = base[kf] + (a - (dp + 1));
}
__global__ void k_tree_split_anc(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ keys2,
                                 const signed char* __restrict__ delta,
                                 const int* __restrict__ base, int n, int n_nodes, int n_split,
                                 int* __restrict__ first, int* __restrict__ n_anc, int* __restrict__ anc, int max_anc,
                                 const int* __restrict__ info, int* __restrict__ poison) {
    split_anc_block(keys, keys2, delta, base, n, n_nodes, n_split, first, n_anc, anc, max_anc, info, poison, -1, blockIdx.x, threadIdx.x);
}

// ---- the two scans of the build in three launches with a FIXED association order: per body k the exclusive
// sum of emit_count (its first node's pre-order index) and the inclusive sum of {m, m x, m y, m z} in f64.
// rocPRIM's decoupled look-back scan combines the partial sums of earlier blocks in whatever grouping their
// completion order allows; f64 addition is not associative, so a centre of mass would now and then differ in its last
// f32 bit from one run to the next (seen: 1 node in 30 000, every few builds).  Here: (1) k_tree_delta_totals computes
// delta/emit_count/the Sum4 terms and each tile's totals (1 024 bodies per tile), (2) k_tree_scan adds up the totals of
// the tiles before its own sequentially (tile order) and scans its tile locally (Hillis-Steele in LDS: a fixed
// pattern).  One launch each -- the separate rocPRIM integer scan (2 launches) and the three-pass Sum4 scan of
// round 1 cost 7 launches of ~5 us with k_tree_delta.
constexpr int kScanThreads = 256, kScanItems = 4, kScanTile = kScanThreads * kScanItems;

__device__ __forceinline__ Sum4 sum4_add(const Sum4& a, const Sum4& b) { return Sum4{a.m + b.m, a.x + b.x, a.y + b.y, a.z + b.z}; }

struct ScanItem { Sum4 s; int c; int w; };   // Sum4 terms, nodes emitted, weight (spatial shards; rides in the padding)
__device__ __forceinline__ ScanItem item_add(const ScanItem& a, const ScanItem& b) { return ScanItem{sum4_add(a.s, b.s), a.c + b.c, a.w + b.w}; }

// inclusive scan of one value per thread over the block (Hillis-Steele in LDS: a fixed pattern)
template <int THREADS>
__device__ __forceinline__ ScanItem block_inclusive_scan(ScanItem v, ScanItem* lds) {
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int off = 1; off < THREADS; off <<= 1) {
        ScanItem add = ScanItem{Sum4{0.0, 0.0, 0.0, 0.0}, 0, 0};
        const bool has = t >= off;
        if (has) add = lds[t - off];
        __syncthreads();
        if (has) { v = item_add(add, v); lds[t] = v; }
        __syncthreads();
    }
    return v;
}

// per sorted body k: delta[k] (common levels with its right neighbour), emit_count[k] (cells it opens + its leaf), the
// Sum4 term; per tile of kScanTile bodies: the totals of both
template <class P4>
__global__ __launch_bounds__(kScanThreads) void k_tree_delta_totals(const unsigned long long* __restrict__ keys,
                                                                     const unsigned long long* __restrict__ keys2,
                                                                     const int* __restrict__ ids, const P4* __restrict__ pos,
                                                                     const int* __restrict__ count, signed char* __restrict__ delta,
                                                                     int* __restrict__ emit_count, Sum4* __restrict__ sums,
                                                                     int* __restrict__ flags, ScanItem* __restrict__ totals,
                                                                     const int* __restrict__ edge, const float4* __restrict__ weight_src,
                                                                     int* __restrict__ weight) {
    __shared__ ScanItem lds[kScanThreads];
    const int n = *count;
    // spatial shards (nbody_let.cpp): this rank's sorted bodies are a contiguous piece of the GLOBAL sorted order, and
    // its first and last body have a neighbour on another rank: edge[0] / edge[1] = levels they share with it (-1: none)
    const int edge_prev = edge ? edge[0] : -1, edge_next = edge ? edge[1] : -1;
    const int k0 = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    ScanItem v = ScanItem{Sum4{0.0, 0.0, 0.0, 0.0}, 0, 0};
#pragma unroll
    for (int q = 0; q < kScanItems; ++q) {
        const int k = k0 + q;
        if (k >= n) break;
        const int d_next = (k + 1 < n) ? common_levels2(keys, keys2, k, k + 1) : edge_next;
        const int d_prev = (k > 0) ? common_levels2(keys, keys2, k - 1, k) : edge_prev;
        if (d_next >= 2 * kLevels) atomicOr(flags, 1);  // two bodies share all 42 levels: too deep for this build
        delta[k] = (signed char)d_next;
        const int ec = max(0, d_next - d_prev) + 1;  // opened cells + the leaf
        emit_count[k] = ec;
        const int id = ids[k];
        const P4 p = pos[id];
        const Sum4 t = Sum4{double(p.w), double(p.w) * double(p.x), double(p.w) * double(p.y), double(p.w) * double(p.z)};
        sums[k] = t;
        int w = 0;
        if (weight_src) { w = body_weight(weight_src[id].w); weight[k] = w; }
        v = item_add(v, ScanItem{t, ec, w});
    }
    v = block_inclusive_scan<kScanThreads>(v, lds);
    if (threadIdx.x == kScanThreads - 1) totals[blockIdx.x] = v;
}

// base[k] = exclusive scan of emit_count, incl[k] = inclusive scan of the Sum4 terms
__global__ __launch_bounds__(kScanThreads) void k_tree_scan(const int* __restrict__ emit_count, const Sum4* __restrict__ sums,
                                                            const int* __restrict__ count, const ScanItem* __restrict__ totals,
                                                            int* __restrict__ base, Sum4* __restrict__ incl, int* __restrict__ out_info,
                                                            int* __restrict__ weight /* in: per body; out: exclusive prefix; may be null */) {
    __shared__ ScanItem lds[kScanThreads];
    __shared__ ScanItem carry_s;
    const int n = *count;
    const int k0 = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    if (blockIdx.x * kScanTile >= n) return;
    // the totals of the tiles before this one: thread t adds its run of consecutive tiles, the block scans the 256
    // partial sums -- a fixed pattern that depends on the tile index only, so every build associates alike (one
    // thread adding 64 totals one after the other was a 16 us chain of dependent loads)
    {
        const int before = int(blockIdx.x);
        const int per = (before + kScanThreads - 1) / kScanThreads;
        ScanItem run = ScanItem{Sum4{0.0, 0.0, 0.0, 0.0}, 0, 0};
        for (int q = 0; q < per; ++q) {
            const int b = int(threadIdx.x) * per + q;
            if (b < before) run = item_add(run, totals[b]);
        }
        const ScanItem all = block_inclusive_scan<kScanThreads>(run, lds);
        if (threadIdx.x == kScanThreads - 1) carry_s = all;
        __syncthreads();
    }
    ScanItem item[kScanItems];
    ScanItem v = ScanItem{Sum4{0.0, 0.0, 0.0, 0.0}, 0, 0};
#pragma unroll
    for (int q = 0; q < kScanItems; ++q) {
        item[q] = (k0 + q < n) ? ScanItem{sums[k0 + q], emit_count[k0 + q], weight ? weight[k0 + q] : 0} : ScanItem{Sum4{0.0, 0.0, 0.0, 0.0}, 0, 0};
        v = item_add(v, item[q]);
    }
    const ScanItem inc = block_inclusive_scan<kScanThreads>(v, lds);
    __syncthreads();
    lds[threadIdx.x] = inc;
    __syncthreads();
    ScanItem run = item_add(carry_s, threadIdx.x > 0 ? lds[threadIdx.x - 1] : ScanItem{Sum4{0.0, 0.0, 0.0, 0.0}, 0, 0});
#pragma unroll
    for (int q = 0; q < kScanItems; ++q) {
        if (k0 + q < n) { base[k0 + q] = run.c; if (weight) weight[k0 + q] = run.w; }
        run = item_add(run, item[q]);
        if (k0 + q < n) incl[k0 + q] = run.s;
        if (k0 + q == n - 1) { out_info[0] = run.c; out_info[2] = n; }   // nodes in all; the live body count rides along (one read-back)
    }
}

// bytes of scratch the passes need for n_cap bodies
size_t sum4_scan_tmp_bytes(size_t n_cap) { return ((n_cap + kScanTile - 1) / kScanTile + 1) * sizeof(ScanItem); }

// scratch at the start of the build workspace: whatever the rocPRIM sort / integer scan or the Sum4
// scan asks for, whichever is largest
size_t scratch_bytes(size_t n_cap) {
    size_t sort_bytes = 0, scan_i = 0;
    unsigned long long* k = nullptr; int* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, k, k, v, v, n_cap, 0, 64, 0);
    (void)rocprim::exclusive_scan(nullptr, scan_i, v, v, 0, n_cap, rocprim::plus<int>(), 0);
    // (at least 4 KB: k_tree_ties keeps its list of big groups there between the sort and the scans)
    return (std::max<size_t>(4096, std::max(sort_bytes, std::max(scan_i, sum4_scan_tmp_bytes(n_cap)))) + 255) / 256 * 256;
}

// ---- sharded runs: the tree is built over the live bodies of ALL segments (every GPU builds the
// same tree from the gathered positions and walks it for its own bodies)
// live bodies of the segments, concatenated in segment order; info[0] = total, info[1] = index of the
// own segment's first body in the concatenation, info[2] = own count
__global__ __launch_bounds__(256) void k_tree_cat(const float4* __restrict__ pos_all, const int* __restrict__ seg_count,
                                                  int n_seg, int seg_cap, int my_seg, float4* __restrict__ pos_cat,
                                                  int* __restrict__ info) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int sgm = idx / seg_cap, j = idx - sgm * seg_cap;
    if (sgm >= n_seg) return;
    int first = 0, own_first = 0, total = 0;
    for (int t = 0; t < n_seg; ++t) {
        if (t == sgm) first = total;
        if (t == my_seg) own_first = total;
        total += seg_count[t];
    }
    if (idx == 0) { info[0] = total; info[1] = own_first; info[2] = seg_count[my_seg]; }
    if (j < seg_count[sgm]) pos_cat[first + j] = pos_all[size_t(sgm) * seg_cap + j];
}

__global__ __launch_bounds__(256) void k_tree_cat64(const double4* __restrict__ pos_all, const int* __restrict__ seg_count,
                                                    int n_seg, int seg_cap, int my_seg, double4* __restrict__ pos_cat,
                                                    int* __restrict__ info) {   // (k_tree_cat for F = f64)
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int sgm = idx / seg_cap, j = idx - sgm * seg_cap;
    if (sgm >= n_seg) return;
    int first = 0, own_first = 0, total = 0;
    for (int t = 0; t < n_seg; ++t) {
        if (t == sgm) first = total;
        if (t == my_seg) own_first = total;
        total += seg_count[t];
    }
    if (idx == 0) { info[0] = total; info[1] = own_first; info[2] = seg_count[my_seg]; }
    if (j < seg_count[sgm]) pos_cat[first + j] = pos_all[size_t(sgm) * seg_cap + j];
}

// tree-order list of all bodies -> flags of the own ones
__global__ __launch_bounds__(256) void k_tree_own_flags(const int* __restrict__ order, const int* __restrict__ info,
                                                        int* __restrict__ flags) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= info[0]) return;
    const int rel = order[k] - info[1];
    flags[k] = (rel >= 0 && rel < info[2]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_tree_own_scatter(const int* __restrict__ order, const int* __restrict__ info,
                                                          const int* __restrict__ flags, const int* __restrict__ base,
                                                          int* __restrict__ own_order) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= info[0]) return;
    if (flags[k]) own_order[base[k]] = order[k] - info[1];  // index inside the own segment
}

}  // namespace

size_t tree_build_workspace_bytes(size_t n_cap) {
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    return scratch_bytes(n_cap) + 3 * al(n_cap * 8) + 5 * al(n_cap * 4) + al(n_cap) + 2 * al(n_cap * sizeof(Sum4)) + 256;
}

namespace {
struct BuildLayout {
    void* tmp; size_t tmp_bytes;
    unsigned long long *keys_in, *keys, *keys2;
    int *ids_in, *ids, *emit_count, *base, *wpre;
    signed char* delta;
    Sum4 *sums, *incl;
    int* counters;   // [1] k_tree_ties: groups listed for the workgroup sort (cleared by k_tree_keys)
};
BuildLayout build_layout(void* workspace, size_t n_cap) {
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    char* p = static_cast<char*>(workspace);
    BuildLayout L;
    L.tmp_bytes = scratch_bytes(n_cap);
    L.tmp = p; p += L.tmp_bytes;
    L.keys_in = reinterpret_cast<unsigned long long*>(p); p += al(n_cap * 8);
    L.keys = reinterpret_cast<unsigned long long*>(p); p += al(n_cap * 8);
    L.keys2 = reinterpret_cast<unsigned long long*>(p); p += al(n_cap * 8);
    L.ids_in = reinterpret_cast<int*>(p); p += al(n_cap * 4);
    L.ids = reinterpret_cast<int*>(p); p += al(n_cap * 4);
    L.emit_count = reinterpret_cast<int*>(p); p += al(n_cap * 4);
    L.base = reinterpret_cast<int*>(p); p += al(n_cap * 4);
    L.wpre = reinterpret_cast<int*>(p); p += al(n_cap * 4);
    L.delta = reinterpret_cast<signed char*>(p); p += al(n_cap);
    L.sums = reinterpret_cast<Sum4*>(p); p += al(n_cap * sizeof(Sum4));
    L.incl = reinterpret_cast<Sum4*>(p); p += al(n_cap * sizeof(Sum4));
    L.counters = reinterpret_cast<int*>(p);   // (the 256 bytes tree_build_workspace_bytes adds at the end)
    return L;
}
}  // namespace

// ---- the build's host side, for F = f32 (float4 bodies, two float4 per node) and F = f64 (double4 bodies, Node64)
namespace {
template <class P4, class Real>
int sort_keys_t(hipStream_t s, const P4* pos, const int* d_count, int n_upper, const Real center[3], Real width, void* workspace, size_t n_cap,
                int* out_info, TreeDevWork* work) {
    const BuildLayout L = build_layout(workspace, n_cap);
    work->keys = L.keys; work->keys2 = L.keys2; work->wpre = L.wpre; work->delta = L.delta; work->base = L.base; work->ids = L.ids; work->incl = L.incl;
    const int n = n_upper;
    if (n <= 0) { (void)hipMemsetAsync(out_info, 0, 2 * sizeof(int), s); return 0; }  // (k_tree_keys clears it otherwise)
    hipLaunchKernelGGL((k_tree_keys<P4, Real>), dim3((n + 255) / 256), dim3(256), 0, s, pos, d_count, n, center[0], center[1], center[2], width,
                       L.keys_in, L.ids_in, out_info, L.counters);
    size_t tb = L.tmp_bytes;
    // (bits [15, 63): the unused tail's keys are all ones and stay behind every real body -- the sort is stable and the
    // tail comes last in the input; k_tree_ties finishes the low bits)
    if (rocprim::radix_sort_pairs(L.tmp, tb, L.keys_in, L.keys, L.ids_in, L.ids, size_t(n), kSortLowBits, 63, s) != hipSuccess) return -1;
    hipLaunchKernelGGL((k_tree_ties<P4, Real>), dim3((n + 255) / 256), dim3(256), 0, s, pos, d_count, center[0], center[1], center[2], width,
                       L.keys, L.keys2, L.ids, out_info + 1, std::max(1, tuning().tree_max_tie));
    return 0;
}
template <class P4>
int scan_sorted_t(hipStream_t s, const P4* pos, const int* d_count, int n_upper, void* workspace, size_t n_cap, int* out_info, const int* edge,
                  const float4* weight_src) {
    const BuildLayout L = build_layout(workspace, n_cap);
    const int n = n_upper;
    if (n <= 0) return 0;
    const int n_tiles = (n + kScanTile - 1) / kScanTile;
    ScanItem* totals = static_cast<ScanItem*>(L.tmp);   // (the sort is done with its scratch)
    hipLaunchKernelGGL((k_tree_delta_totals<P4>), dim3(n_tiles), dim3(kScanThreads), 0, s, L.keys, L.keys2, L.ids, pos, d_count, L.delta,
                       L.emit_count, L.sums, out_info + 1, totals, edge, weight_src, L.wpre);
    hipLaunchKernelGGL(k_tree_scan, dim3(n_tiles), dim3(kScanThreads), 0, s, L.emit_count, L.sums, d_count, totals, L.base, L.incl, out_info,
                       weight_src ? L.wpre : nullptr);
    return 0;
}
template <class P4, class Real, class NodeT>
int emit_nodes_t(hipStream_t s, const P4* pos, const int* d_count, Real width, void* workspace, size_t n_cap, NodeT* nodes, int node_cap,
                 int slice_cap, int* order, int* out_info, int want_hot, const int* edge, const int* node_offset, int* parent,
                 unsigned char* depth, const TreeSplitReq* split = nullptr) {
    const BuildLayout L = build_layout(workspace, n_cap);
    // one thread per node; their number is known on the device only, so one per node the slice can have
    // (threads beyond the tree leave at once; a tree beyond the array sets flag 2 and the caller grows it)
    const int emit_blocks = (std::max(1, slice_cap) + 255) / 256;
    const TreeSplitReq req = split ? *split : TreeSplitReq{};
    hipLaunchKernelGGL((k_tree_emit<P4, Real, NodeT>), dim3(emit_blocks + req.n_split), dim3(256), 0, s, L.keys, L.keys2, L.ids, pos,
                       d_count, L.delta, L.base, L.incl, width, nodes, node_cap, order, out_info, want_hot, edge, node_offset, parent, depth, req,
                       emit_blocks);
    return 0;
}
}  // namespace

// First half of the build: keys and the sort.  work->keys / work->ids = the sorted (key, body) pairs.
int tree_sort_keys(hipStream_t s, const float4* pos, const int* d_count, int n_upper, const float center[3], float width,
                   void* workspace, size_t n_cap, int* out_info, TreeDevWork* work) {
    return sort_keys_t<float4, float>(s, pos, d_count, n_upper, center, width, workspace, n_cap, out_info, work);
}

// Second half: delta and the scans (out_info[0] = nodes in all, out_info[2] = bodies), then the emit.  edge (device, 2
// ints, may be null): levels the first / last sorted body shares with its neighbour on another rank (spatial shards), see
// k_tree_delta_totals.
int tree_scan_sorted(hipStream_t s, const float4* pos, const int* d_count, int n_upper, void* workspace, size_t n_cap, int* out_info,
                     const int* edge, const float4* weight_src) {
    return scan_sorted_t<float4>(s, pos, d_count, n_upper, workspace, n_cap, out_info, edge, weight_src);
}
// node_offset (device, may be null): the slice is written at nodes[*node_offset ..] with its links shifted; parent /
// depth (may be null): per node of the slice, see k_tree_emit
int tree_emit_nodes(hipStream_t s, const float4* pos, const int* d_count, int n_upper, float width, void* workspace, size_t n_cap,
                    float4* nodes, int node_cap, int slice_cap, int* order, int* out_info, int want_hot, const int* edge,
                    const int* node_offset, int* parent, unsigned char* depth) {
    (void)n_upper;
    return emit_nodes_t<float4, float, float4>(s, pos, d_count, width, workspace, n_cap, nodes, node_cap, slice_cap, order, out_info, want_hot, edge,
                                               node_offset, parent, depth);
}
int tree_emit_sorted(hipStream_t s, const float4* pos, const int* d_count, int n_upper, float width, void* workspace, size_t n_cap,
                     float4* nodes, int node_cap, int* order, int* out_info, int want_hot, const int* edge, const TreeSplitReq* split) {
    if (tree_scan_sorted(s, pos, d_count, n_upper, workspace, n_cap, out_info, edge) != 0) return -1;
    (void)n_upper;
    return emit_nodes_t<float4, float, float4>(s, pos, d_count, width, workspace, n_cap, nodes, node_cap, node_cap, order, out_info, want_hot, edge,
                                               nullptr, nullptr, nullptr, split);
}

// Enqueues the whole build on `s`.  out_info (device, 3 ints): [0] = node count, [1] = flags, [2] = bodies in the tree
// (1: deeper than 42 levels, 2: node_cap too small).  The caller reads it back before the walk.
int build_octree_device(hipStream_t s, const float4* pos, const int* d_count, int n_upper, const float center[3],
                        float width, void* workspace, size_t n_cap, float4* nodes, int node_cap, int* order,
                        int* out_info, TreeDevWork* work, int want_hot, const TreeSplitReq* split) {
    if (tree_sort_keys(s, pos, d_count, n_upper, center, width, workspace, n_cap, out_info, work) != 0) return -1;
    return tree_emit_sorted(s, pos, d_count, n_upper, width, workspace, n_cap, nodes, node_cap, order, out_info, want_hot, nullptr, split);
}
// The same for F = f64: keys from the reference's recurrences in double, centres of mass from the same f64 prefix sums
// (against the reference's sequential f64 folds they differ in the last bits only), 64-byte node records.
int build_octree_device_f64(hipStream_t s, const double4* pos, const int* d_count, int n_upper, const double center[3], double width,
                            void* workspace, size_t n_cap, nbody64::Node64* nodes, int node_cap, int* order, int* out_info,
                            TreeDevWork* work, const TreeSplitReq* split) {
    if (sort_keys_t<double4, double>(s, pos, d_count, n_upper, center, width, workspace, n_cap, out_info, work) != 0) return -1;
    if (scan_sorted_t<double4>(s, pos, d_count, n_upper, workspace, n_cap, out_info, nullptr, nullptr) != 0) return -1;
    return emit_nodes_t<double4, double, nbody64::Node64>(s, pos, d_count, width, workspace, n_cap, nodes, node_cap, node_cap, order, out_info, 0,
                                                          nullptr, nullptr, nullptr, nullptr, split);
}

void launch_tree_split_anc(hipStream_t s, const TreeDevWork& work, int n, int n_nodes, int n_split, int* first,
                           int* n_anc, int* anc, int max_anc, const int* info, int* poison) {
    hipLaunchKernelGGL(k_tree_split_anc, dim3(n_split), dim3(64), 0, s, work.keys, work.keys2, work.delta, work.base, n, n_nodes,
                       n_split, first, n_anc, anc, max_anc, info, poison);
}

#ifdef NBODY_TUNING   // (the cooperative block walk is one of the experimental walks: tuning build only)
// ---- level-order copy of the tree for the cooperative block walk (kernels_bh.hip k_bh_walk_block): the nodes sorted
// by depth, ties in pre-order, so that the children of a node are consecutive records (all descendants of a node at
// one depth lie between its pre-order bounds).  Record: {com, mass | w^2, pre-order index, pre-order skip link,
// position of the first child | last-sibling flag in bit 31}.  Works on any pre-order array (host or device build):
// the depth comes from the width, which halves exactly per level (shared.rs:256-272).
namespace {

__global__ __launch_bounds__(256) void k_bfs_keys(const float4* __restrict__ nodes, int n_nodes, unsigned char* __restrict__ depth,
                                                  unsigned int* __restrict__ keys, int* __restrict__ vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes) return;
    // w^2 = W^2 * 4^-d exactly: the exponent fields differ by 2d
    const int d = (__float_as_int(nodes[1].x) - __float_as_int(nodes[2 * i + 1].x)) >> 24;
    const int dc = min(max(d, 0), 63);
    depth[i] = (unsigned char)dc;
    keys[i] = (unsigned int)dc;
    vals[i] = i;
}

__global__ __launch_bounds__(256) void k_bfs_inverse(const int* __restrict__ perm, int n_nodes, int* __restrict__ pos) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < n_nodes) pos[perm[p]] = p;
}

__global__ __launch_bounds__(256) void k_bfs_records(const float4* __restrict__ nodes, int n_nodes, const int* __restrict__ perm,
                                                     const int* __restrict__ pos, const unsigned char* __restrict__ depth,
                                                     float4* __restrict__ out) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_nodes) return;
    const int i = perm[p];
    const float4 a = nodes[2 * i], b = nodes[2 * i + 1];
    const int skip = __float_as_int(b.y);
    const int child = (skip > i + 1) ? pos[i + 1] : 0;                        // the first child is the next node in pre-order
    const bool last = (skip >= n_nodes) || (depth[skip] < depth[i]);          // behind the subtree: a sibling, or an ancestor's sibling
    out[2 * p] = a;
    out[2 * p + 1] = make_float4(b.x, __int_as_float(i), __int_as_float(skip), __int_as_float(child | (last ? int(0x80000000u) : 0)));
}

size_t bfs_sort_bytes(size_t n_cap) {
    size_t b = 0;
    unsigned int* k = nullptr; int* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, b, k, k, v, v, n_cap, 0, 6, 0);
    return (b + 255) / 256 * 256;
}

}  // namespace

size_t bfs_workspace_bytes(size_t n_cap) {
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    return bfs_sort_bytes(n_cap) + 2 * al(n_cap * 4) + 2 * al(n_cap * 4) + al(n_cap * 4) + al(n_cap) + 256;
}

// enqueues the level-order copy of `nodes` (pre-order, n_nodes records) into `out`; 0 on success
int build_bfs_layout(hipStream_t s, const float4* nodes, int n_nodes, void* workspace, size_t n_cap, float4* out) {
    if (n_nodes <= 0) return 0;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    char* p = static_cast<char*>(workspace);
    const size_t sort_bytes = bfs_sort_bytes(n_cap);
    void* tmp = p; p += sort_bytes;
    auto* keys_in = reinterpret_cast<unsigned int*>(p); p += al(n_cap * 4);
    auto* keys = reinterpret_cast<unsigned int*>(p); p += al(n_cap * 4);
    auto* vals_in = reinterpret_cast<int*>(p); p += al(n_cap * 4);
    auto* perm = reinterpret_cast<int*>(p); p += al(n_cap * 4);
    auto* pos = reinterpret_cast<int*>(p); p += al(n_cap * 4);
    auto* depth = reinterpret_cast<unsigned char*>(p);
    const dim3 grid((n_nodes + 255) / 256), block(256);
    hipLaunchKernelGGL(k_bfs_keys, grid, block, 0, s, nodes, n_nodes, depth, keys_in, vals_in);
    size_t tb = sort_bytes;
    if (rocprim::radix_sort_pairs(tmp, tb, keys_in, keys, vals_in, perm, size_t(n_nodes), 0, 6, s) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_bfs_inverse, grid, block, 0, s, perm, n_nodes, pos);
    hipLaunchKernelGGL(k_bfs_records, grid, block, 0, s, nodes, n_nodes, perm, pos, depth, out);
    return 0;
}

#endif  // NBODY_TUNING

// bytes at the start of the build workspace that rocPRIM uses as scratch (free between builds)
size_t tree_build_tmp_bytes(size_t n_cap) { return scratch_bytes(n_cap); }

// sharded runs: bytes of the side buffer (concatenated positions, own-order flags/offsets/list, info)
size_t tree_cat_bytes(size_t n_cap) {
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    return al(n_cap * sizeof(float4)) + 3 * al(n_cap * 4) + 256;
}

TreeCat tree_cat_layout(void* buf, size_t n_cap) {
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    char* p = static_cast<char*>(buf);
    TreeCat c;
    c.pos = reinterpret_cast<float4*>(p); p += al(n_cap * sizeof(float4));
    c.flags = reinterpret_cast<int*>(p); p += al(n_cap * 4);
    c.base = reinterpret_cast<int*>(p); p += al(n_cap * 4);
    c.own_order = reinterpret_cast<int*>(p); p += al(n_cap * 4);
    c.info = reinterpret_cast<int*>(p);
    return c;
}

// F = f64: the side buffer holds double4 positions (tree_cat_bytes64); flags / base / own_order / info as in TreeCat (pos unused)
size_t tree_cat_bytes64(size_t n_cap) {
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    return al(n_cap * sizeof(double4)) + 3 * al(n_cap * 4) + 256;
}
TreeCat tree_cat_layout64(void* buf, size_t n_cap, double4** pos_cat) {
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    char* p = static_cast<char*>(buf);
    TreeCat c;
    *pos_cat = reinterpret_cast<double4*>(p); p += al(n_cap * sizeof(double4));
    c.pos = nullptr;
    c.flags = reinterpret_cast<int*>(p); p += al(n_cap * 4);
    c.base = reinterpret_cast<int*>(p); p += al(n_cap * 4);
    c.own_order = reinterpret_cast<int*>(p); p += al(n_cap * 4);
    c.info = reinterpret_cast<int*>(p);
    return c;
}
void launch_tree_cat64(hipStream_t s, const double4* pos_all, const int* seg_count, int n_seg, int seg_cap, int my_seg, double4* pos_cat, int* info) {
    const int slots = n_seg * seg_cap;
    hipLaunchKernelGGL(k_tree_cat64, dim3((slots + 255) / 256), dim3(256), 0, s, pos_all, seg_count, n_seg, seg_cap, my_seg, pos_cat, info);
}

void launch_tree_cat(hipStream_t s, const Shard& sh, const TreeCat& c) {
    const int slots = sh.n_seg * sh.seg_cap;
    hipLaunchKernelGGL(k_tree_cat, dim3((slots + 255) / 256), dim3(256), 0, s, sh.pos_all, sh.seg_count, sh.n_seg, sh.seg_cap,
                       sh.my_seg, c.pos, c.info);
}

// own bodies in tree order, as indices into the own segment (tmp: the build's workspace, free again)
int launch_tree_own_order(hipStream_t s, const int* order, const TreeCat& c, int n_total_upper, void* tmp, size_t tmp_bytes) {
    if (n_total_upper <= 0) return 0;
    const dim3 grid((n_total_upper + 255) / 256), block(256);
    hipLaunchKernelGGL(k_tree_own_flags, grid, block, 0, s, order, c.info, c.flags);
    size_t tb = tmp_bytes;
    // (entries beyond the live total are never read back: flags there may be stale, their offsets unused)
    if (rocprim::exclusive_scan(tmp, tb, c.flags, c.base, 0, size_t(n_total_upper), rocprim::plus<int>(), s) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_tree_own_scatter, grid, block, 0, s, order, c.info, c.flags, c.base, c.own_order);
    return 0;
}

}  // namespace nbody
