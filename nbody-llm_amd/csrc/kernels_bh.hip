// kernels_bh.hip -- K5: Barnes-Hut tree walk, BarnesHutSimulation::calc_force
// (src/manual/barnes_hut.rs:185-203), over the linearised octree built by octree_host.cpp.
//
// The octree is stored in depth-first pre-order with children in orthant order 0..7 -- the order
// the reference's recursion visits them -- and every node carries `skip`, the index of the first
// node after its subtree.  The recursion then becomes a loop with no stack at all:
//     accept (w^2 < theta2 * r^2)  -> add the monopole, jump to skip
//     otherwise                    -> step to i + 1 (the first child; for a leaf i + 1 == skip,
//                                     which is the reference's "leaf that fails the test adds 0")
// Every lane evaluates exactly the opening tests the reference evaluates for its body, in the
// same order and with the same rounding (r2 = (x*x + y*y) + z*z, no contraction), so the
// accepted-node and visited-node counts equal the reference's.  The fast walk (k_bh_walk) adds the
// accepted monopoles into one running sum per lane; the reference nests the sums per tree level, so
// fast math agrees to rounding (tolerance in tests/test_bh_gpu.py).  Strict math runs
// k_bh_walk_nested, which reproduces the nesting: accelerations bit-equal to the oracle's.
//
// Bodies are walked in tree (depth-first leaf) order, so the 64 lanes of a wave hold spatial
// neighbours and follow nearly the same path: their node reads coalesce into a few 32-byte
// sectors that stay in L1/L2 (the whole node array is 32 B x n_nodes, ~4 MB at N = 65 536).
#include "kernels.h"

namespace nbody {

constexpr int kWalkBlock = 64;   // one wave per workgroup: four body groups from different parts of the tree per CU (0.33 ms against 0.36 with 256)
constexpr unsigned kCounterSlots = NBODY_WALK_COUNTER_SLOTS;  // {accepted, visited} pairs the waves' counts are spread over

typedef float f32x8 __attribute__((ext_vector_type(8)));
struct alignas(32) NodeDev { float4 a; float4 b; };  // {com, mass}, {width^2, skip bits, width, leaf body}

// The node index range [0, n_nodes) can be cut into n_seg contiguous segments walked by different
// waves (more waves in flight: at N = 65 536 one wave per 64 bodies is only one wave per SIMD and the
// walk is bound by the latency of its dependent node loads).  A body's walk enters segment k at the
// first node >= first[k] that it would visit: to know, replay the opening tests of the ancestors of
// node first[k] (root first; the host lists them, at most NBODY_MAX_TREE_DEPTH): an accepted
// ancestor's skip link is where the walk resumes.  Ancestors are only tested here -- they are
// counted and accumulated by the segment that contains them -- so every (body, node) pair is
// evaluated by exactly one segment and the counters stay exact.
struct WalkSplit {
    int n_seg;
    const int* first;        // [n_seg + 1] node index where each segment starts; first[n_seg] = n_nodes
    const int* anc;          // [n_seg][kMaxAnc] ancestors of first[k], root first
    const int* n_anc;        // [n_seg]
    float4* planes;          // [n_seg][plane_stride] partial accelerations (n_seg > 1), indexed by the body's place in `order`
                             // (tree order): the walk's lanes and the reduction's both touch consecutive entries
    size_t plane_stride;
    int diag_first;          // k_bh_walk: segments of a body group in order of distance from its own place in the tree
    const int* poison;       // unsynchronised steps: != 0 -> do nothing (Shard::poison); may be null
    const int* n_order_dev;  // unsynchronised steps: the live number of bodies to walk (the host's is an upper bound); may be null
    int store_work;          // k_bh_walk, one segment: the body's visit count goes to acc.w (spatial shards balance by it)
    int xcd_blocks;          // k_bh_walk_duo: gridDim.x / 8 when the lane groups are dealt to the XCDs in eighths of the tree order, else 0
};
constexpr int kMaxAnc = 192;

template <bool DIRECT = false>
__device__ __forceinline__ int walk_entry(const NodeDev* __restrict__ nodes, const WalkSplit& sp, int seg,
                                          const float4 p, float theta2) {
    const int s0 = sp.first[seg];
    const int na = sp.n_anc[seg];
    for (int k = 0; k < na; ++k) {
        const int j = sp.anc[seg * kMaxAnc + k];
        const float4 A = nodes[j].a;
        const float4 B = nodes[j].b;
        const float rx = A.x - p.x, ry = A.y - p.y, rz = A.z - p.z;
        const float r2 = (rx * rx + ry * ry) + rz * rz;
        if (DIRECT && r2 < 1e-10f) return __float_as_int(B.y);  // NBODY_LEAF_DIRECT: skipped whole
        if (B.x < theta2 * r2) return __float_as_int(B.y);  // accepted: the walk resumes after its subtree
    }
    return s0;  // every ancestor was opened: the walk arrives at first[seg] itself
}

#ifdef NBODY_TUNING
__device__ unsigned long long nbody_bh_stamps[3 * 65536];  // diagnostic build only (DBG): per wave start, end (100 MHz ticks), iterations
#endif

// DIRECT = NBODY_LEAF_DIRECT: the walk of src/llm/barnes_hut.rs:915-997 on the same tree (see nbody_hip.h)
template <bool FAST, bool DIRECT = false, bool DBG = false, int BLOCK = kWalkBlock>
__global__ __launch_bounds__(BLOCK) void k_bh_walk(const NodeDev* __restrict__ nodes, int n_nodes,
                                                        const int* __restrict__ order, int n_order,
                                                        const float4* __restrict__ own_pos, float4* __restrict__ acc,
                                                        float g, float eps2, float theta2,
                                                        unsigned long long* __restrict__ counters, WalkSplit split) {
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    // Which of the K segments: the launch lasts as long as its slowest wave, and a body group's long walks are
    // in the segments around its own place in the tree (that is where cells are opened down to the leaves).
    // Bodies are in tree order, so group x of X sits near node x/X * n_nodes: the segments are taken by
    // distance from that "diagonal" -- blockIdx.y = 0 is the group's own segment, then +1, -1, +2, ... -- and
    // the dispatcher, which hands out workgroups in blockIdx order (x fastest), starts the heavy ones first.
    int seg = blockIdx.y;
    if (split.diag_first) {
        const int K = gridDim.y;
        const int diag = int((long long)blockIdx.x * K / gridDim.x);
        const int kk = blockIdx.y;
        const int off = (kk & 1) ? (kk + 1) / 2 : -(kk / 2);
        seg = ((diag + off) % K + K) % K;
    }
    if (split.poison && *split.poison) return;
    if (split.n_order_dev) n_order = min(n_order, *split.n_order_dev);
    const int s1 = split.first[seg + 1];
    unsigned int n_acc = 0, n_vis = 0;
    [[maybe_unused]] unsigned long long r_beg = 0;
    if (DBG) r_beg = __builtin_amdgcn_s_memrealtime();
    if (t < n_order) {
        const int b = order[t];
        const float4 p = own_pos[b];
        float ax = 0.f, ay = 0.f, az = 0.f;
        int i = walk_entry<DIRECT>(nodes, split, seg, p, theta2);  // first node >= s0 this body's walk visits
        while (i < s1) {
            const float4 A = nodes[i].a;
            // of the record's second half the walk needs {w^2, skip link} only: an 8-byte load (24 instead of 32
            // bytes per visit through the L1's 64 B/clk return path)
            const float2 B = *reinterpret_cast<const float2*>(&nodes[i].b);
            // keep both loads whole and ahead of the branch: left alone the compiler narrows them
            // to {x,y,z} + {w^2} and fetches mass and skip link in a second, dependent round trip under
            // the accept branch (4 L1 accesses and two load latencies per accepted visit)
            asm volatile("" :: "v"(A.w), "v"(B.y));
            const float rx = A.x - p.x, ry = A.y - p.y, rz = A.z - p.z;        // :190
            const float r2 = (rx * rx + ry * ry) + rz * rz;                     // :191
            ++n_vis;
            if (DIRECT) {
                const int skip = __float_as_int(B.y);
                if (r2 < 1e-10f) { i = skip; continue; }                        // llm :933-935 (the body's own leaf: r2 = 0)
                if (B.x < theta2 * r2 || skip == i + 1) {                       // llm :938 accepted cell, :958-972 leaf
                    float k;
                    if (FAST) {
                        const float rinv = __builtin_amdgcn_rsqf(r2 + eps2);
                        k = (g * A.w) * ((rinv * rinv) * rinv);
                    } else {
                        const float inv_r = 1.0f / __builtin_sqrtf(r2 + eps2);  // llm :942
                        const float inv_r3 = inv_r * inv_r * inv_r;             // llm :944
                        k = g * A.w * inv_r3;                                   // llm :947
                    }
                    ax += rx * k; ay += ry * k; az += rz * k;                   // llm :950-952: one running sum
                    ++n_acc;
                    i = skip;
                } else {
                    i = i + 1;
                }
                continue;
            }
            if (B.x < theta2 * r2) {                                            // :192
                float k;
                if (FAST) {
                    const float rinv = __builtin_amdgcn_rsqf(r2 + eps2);
                    k = (g * A.w) * ((rinv * rinv) * rinv);
                } else {
                    const float r_dist = __builtin_sqrtf(r2 + eps2);                 // :193
                    const float r_cubed = r_dist * r_dist * r_dist;             // :194
                    k = ((g * A.w) / r_cubed);                            // :195
                }
                ax += rx * k; ay += ry * k; az += rz * k;
                ++n_acc;
                i = __float_as_int(B.y);
            } else {
                i = i + 1;
            }
        }
        *(split.n_seg > 1 ? split.planes + size_t(seg) * split.plane_stride + t : acc + b) =
            make_float4(ax, ay, az, split.store_work ? float(n_vis) : 0.f);  // :260
    }
#ifdef NBODY_TUNING
    if (DBG) {
        unsigned int it = n_vis;
        for (int off = 32; off > 0; off >>= 1) it = max(it, (unsigned int)__shfl_down(it, off));
        const int w = (blockIdx.y * gridDim.x + blockIdx.x) * (BLOCK / 64) + (threadIdx.x >> 6);
        if ((threadIdx.x & 63) == 0 && w < 16384) {
            const unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
            const unsigned xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
            nbody_bh_stamps[3 * w] = r_beg; nbody_bh_stamps[3 * w + 1] = __builtin_amdgcn_s_memrealtime();
            nbody_bh_stamps[3 * w + 2] = (unsigned long long)it | ((unsigned long long)hw << 20) | ((unsigned long long)(xcc & 0xF) << 52);
        }
    }
#endif
    // one atomic pair per wave
    for (int off = 32; off > 0; off >>= 1) {
        n_acc += __shfl_down(n_acc, off);
        n_vis += __shfl_down(n_vis, off);
    }
    if ((threadIdx.x & 63) == 0 && counters) {
        // one atomic pair per wave, spread over kCounterSlots address pairs: 16 384 atomics on ONE address
        // pair serialise in L2 at ~13 ns each (0.21 ms per walk at 8 segments, measured with theta2 = 1e9)
        const unsigned slot = (blockIdx.x + blockIdx.y * gridDim.x) & (kCounterSlots - 1);
        atomicAdd(&counters[2 * slot], (unsigned long long)n_acc);
        atomicAdd(&counters[2 * slot + 1], (unsigned long long)n_vis);
    }
}

// Two bodies per lane (round 3).  What bounds k_bh_walk is the L1's address rate for divergent gathers, and
// bodies that are neighbours in tree order visit almost the same nodes.  Here a lane walks TWO consecutive bodies of the
// tree order in lockstep: each has its own next index, the lane visits the smaller one, fetches that record ONCE and
// evaluates it for whichever of the two is due there.  Per body these are exactly the opening tests, in exactly the
// order, of its own walk (same sums, same counters); per lane the loads are the UNION of the two sequences instead of
// their sum.
template <bool FAST, bool DIRECT>
__device__ __forceinline__ int duo_visit(const float4 A, const float2 B, int i, const float4 p, float g, float eps2, float theta2,
                                         float& ax, float& ay, float& az, unsigned int& n_acc, unsigned int& n_vis) {
    const float rx = A.x - p.x, ry = A.y - p.y, rz = A.z - p.z;        // :190
    const float r2 = (rx * rx + ry * ry) + rz * rz;                     // :191
    const int skip = __float_as_int(B.y);
    ++n_vis;
    if (DIRECT) {
        if (r2 < 1e-10f) return skip;                                   // llm :933-935
        if (B.x < theta2 * r2 || skip == i + 1) {                       // llm :938, :958-972
            float k;
            if (FAST) { const float rinv = __builtin_amdgcn_rsqf(r2 + eps2); k = (g * A.w) * ((rinv * rinv) * rinv); }
            else { const float inv_r = 1.0f / __builtin_sqrtf(r2 + eps2); const float inv_r3 = inv_r * inv_r * inv_r; k = g * A.w * inv_r3; }
            ax += rx * k; ay += ry * k; az += rz * k;
            ++n_acc;
            return skip;
        }
        return i + 1;
    }
    if (B.x < theta2 * r2) {                                            // :192
        float k;
        if (FAST) { const float rinv = __builtin_amdgcn_rsqf(r2 + eps2); k = (g * A.w) * ((rinv * rinv) * rinv); }
        else { const float r_dist = __builtin_sqrtf(r2 + eps2); const float r_cubed = r_dist * r_dist * r_dist; k = ((g * A.w) / r_cubed); }
        ax += rx * k; ay += ry * k; az += rz * k;
        ++n_acc;
        return skip;
    }
    return i + 1;
}
template <bool FAST, bool DIRECT, int BLOCK, int BPL>
__global__ __launch_bounds__(BLOCK) void k_bh_walk_duo(const NodeDev* __restrict__ nodes, int n_nodes, const int* __restrict__ order, int n_order,
                                                       const float4* __restrict__ own_pos, float4* __restrict__ acc, float g, float eps2,
                                                       float theta2, unsigned long long* __restrict__ counters, WalkSplit split) {
    // Workgroups go to the eight XCDs round-robin by linear id (gridDim.x is a multiple of 8 here): XCD j gets the j-th EIGHTH
    // of the tree order -- a region of space -- instead of every eighth lane group, so that what its L2 holds of the deep
    // nodes is what its next workgroups ask for (Tuning::bh_walk_xcd = 0: the plain order; 2^18 bodies 0.85 -> 0.76 ms, 2^22 10.8 -> 9.5 ms)
    const int bx = split.xcd_blocks ? int(blockIdx.x % 8) * split.xcd_blocks + int(blockIdx.x / 8) : int(blockIdx.x);
    const int t = bx * BLOCK + threadIdx.x;   // the bodies at places BPL t .. BPL t + BPL - 1 of the tree order
    int seg = blockIdx.y;
    if (split.diag_first) {
        const int K = gridDim.y;
        const int diag = int((long long)bx * K / gridDim.x);
        const int kk = blockIdx.y;
        const int off = (kk & 1) ? (kk + 1) / 2 : -(kk / 2);
        seg = ((diag + off) % K + K) % K;
    }
    if (split.poison && *split.poison) return;
    if (split.n_order_dev) n_order = min(n_order, *split.n_order_dev);
    const int s1 = split.first[seg + 1];
    unsigned int n_acc = 0, n_vis = 0;
    const int t0 = BPL * t;
    if (t0 < n_order) {
        float4 p[BPL];
        float ax[BPL], ay[BPL], az[BPL];
        unsigned int v[BPL];
        int nx[BPL], body[BPL];
        int i = s1;
#pragma unroll
        for (int q = 0; q < BPL; ++q) {
            const bool live = t0 + q < n_order;
            body[q] = order[live ? t0 + q : t0];
            p[q] = own_pos[body[q]];
            ax[q] = ay[q] = az[q] = 0.f;
            v[q] = 0;
            nx[q] = live ? walk_entry<DIRECT>(nodes, split, seg, p[q], theta2) : s1;
            i = min(i, nx[q]);
        }
        while (i < s1) {
            const float4 A = nodes[i].a;
            const float2 B = *reinterpret_cast<const float2*>(&nodes[i].b);
            asm volatile("" :: "v"(A.w), "v"(B.y));   // (both loads whole and ahead of the branches: see k_bh_walk)
            int nxt = s1;
#pragma unroll
            for (int q = 0; q < BPL; ++q) {
                if (nx[q] == i) nx[q] = duo_visit<FAST, DIRECT>(A, B, i, p[q], g, eps2, theta2, ax[q], ay[q], az[q], n_acc, v[q]);
                nxt = min(nxt, nx[q]);
            }
            i = nxt;
        }
#pragma unroll
        for (int q = 0; q < BPL; ++q) {
            n_vis += v[q];
            if (t0 + q < n_order)
                *(split.n_seg > 1 ? split.planes + size_t(seg) * split.plane_stride + (t0 + q) : acc + body[q]) =
                    make_float4(ax[q], ay[q], az[q], split.store_work ? float(v[q]) : 0.f);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        n_acc += __shfl_down(n_acc, off);
        n_vis += __shfl_down(n_vis, off);
    }
    if ((threadIdx.x & 63) == 0 && counters) {
        const unsigned slot = (blockIdx.x + blockIdx.y * gridDim.x) & (kCounterSlots - 1);
        atomicAdd(&counters[2 * slot], (unsigned long long)n_acc);
        atomicAdd(&counters[2 * slot + 1], (unsigned long long)n_vis);
    }
}

#ifdef NBODY_TUNING   // ---- experimental walks (variants 1 and 2): measured, slower, kept reproducible in the tuning build only
// Wave-cooperative form of the same walk.  The 64 lanes of a wave hold 64 neighbouring bodies
// (tree order) and step through the UNION of their node sequences together: the node index is
// wave-uniform, so the 32-byte node record arrives by scalar load in SGPRs (no divergent gather),
// and each lane only remembers `resume`, the index at which it becomes interested again after
// accepting a node (= that node's skip link).  A lane evaluates node i iff i >= resume, i.e. iff
// its own sequential walk would visit it; the wave steps to i + 1 while any lane still wants the
// children, else to the skip link.  Per lane the accepted nodes, their order and the counters are
// exactly those of k_bh_walk (and of the reference recursion); only the memory access pattern
// changes.  Measured at N = 65 536, theta = 0.5 with the node range split 8 ways: 0.60 ms against 0.53
// for the per-lane walk -- the union of 64 neighbours' node sequences is several times longer than one
// body's, which costs more than the divergent gathers it saves (groups of 4/8/16/32 lanes were tried
// too: 0.60/0.63/0.68/0.77 ms).  Kept as a selectable variant (tuning().bh_walk_variant = 1).
template <bool FAST>
__global__ __launch_bounds__(kWalkBlock) void k_bh_walk_wave(const NodeDev* __restrict__ nodes, int n_nodes,
                                                             const int* __restrict__ order, int n_order,
                                                             const float4* __restrict__ own_pos,
                                                             float4* __restrict__ acc, float g, float eps2,
                                                             float theta2, unsigned long long* __restrict__ counters,
                                                             WalkSplit split) {
    const int t = blockIdx.x * kWalkBlock + threadIdx.x;
    const int seg = blockIdx.y;
    const int s0 = split.first[seg], s1 = split.first[seg + 1];
    const bool live = t < n_order;
    const int b = live ? order[t] : 0;
    const float4 p = live ? own_pos[b] : make_float4(0.f, 0.f, 0.f, 0.f);
    float ax = 0.f, ay = 0.f, az = 0.f;
    unsigned int n_acc = 0, n_vis = 0;
    // dead lanes never look at a node; live ones start where their own walk enters this segment
    int resume = live ? walk_entry(nodes, split, seg, p, theta2) : 0x7fffffff;
    int i = s0;                          // wave-uniform node index
    while (i < s1) {
        // uniform address: ONE 32-byte scalar load (left to itself the compiler fetches the mass in a
        // third, dependent s_load under the accept branch)
        // uniform address: ONE 32-byte scalar load (left to itself the compiler fetches the mass in a
        // third, dependent s_load under the accept branch)
        f32x8 rec;
        asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rec) : "s"(nodes + i) : "memory");
        const float4 A = make_float4(rec[0], rec[1], rec[2], rec[3]);
        const float4 B = make_float4(rec[4], rec[5], rec[6], rec[7]);
        const bool active = i >= resume;
        const float rx = A.x - p.x, ry = A.y - p.y, rz = A.z - p.z;        // :190
        const float r2 = (rx * rx + ry * ry) + rz * rz;                     // :191
        const bool accept = active && (B.x < theta2 * r2);                  // :192
        n_vis += active ? 1u : 0u;
        if (accept) {
            float k;
            if (FAST) {
                const float rinv = __builtin_amdgcn_rsqf(r2 + eps2);
                k = (g * A.w) * ((rinv * rinv) * rinv);
            } else {
                const float r_dist = __builtin_sqrtf(r2 + eps2);            // :193
                const float r_cubed = r_dist * r_dist * r_dist;             // :194
                k = ((g * A.w) / r_cubed);                                  // :195
            }
            ax += rx * k; ay += ry * k; az += rz * k;
            ++n_acc;
            resume = __float_as_int(B.y);
        }
        const bool wants_children = active && !accept;
        i = __builtin_amdgcn_readfirstlane(__ballot(wants_children) != 0ull ? i + 1 : __float_as_int(B.y));
    }
    if (live) *(split.n_seg > 1 ? split.planes + size_t(seg) * split.plane_stride + t : acc + b) = make_float4(ax, ay, az, 0.f);  // :260
    for (int off = 32; off > 0; off >>= 1) {
        n_acc += __shfl_down(n_acc, off);
        n_vis += __shfl_down(n_vis, off);
    }
    if ((threadIdx.x & 63) == 0 && counters) {
        // one atomic pair per wave, spread over kCounterSlots address pairs: 16 384 atomics on ONE address
        // pair serialise in L2 at ~13 ns each (0.21 ms per walk at 8 segments, measured with theta2 = 1e9)
        const unsigned slot = (blockIdx.x + blockIdx.y * gridDim.x) & (kCounterSlots - 1);
        atomicAdd(&counters[2 * slot], (unsigned long long)n_acc);
        atomicAdd(&counters[2 * slot + 1], (unsigned long long)n_vis);
    }
}


// Variant 2: TWO lanes per body.  A visit needs the node's 32-byte record; one lane fetches it as two
// 16-byte loads (two L1 tag look-ups per visit, which is what bounds k_bh_walk: 0.87 cache accesses
// per cycle per CU), a pair of neighbouring lanes fetches it as the two halves of ONE contiguous
// 32-byte request (measured: 1.0e8 cache-line accesses per launch against 1.5e8).  The even lane holds {com, mass}, takes
// width^2 and the skip link from its odd neighbour by DPP, walks, and hands the next node index back;
// the odd lane executes the same instructions on meaningless values (its branch outcomes are never
// used: every DPP sits outside the divergent region, where both lanes of a pair are active together).
// Result at N = 65 536: 0.39 ms with the node range split 4 ways, the same as the per-lane walk's
// 0.38 ms (8 ways): the L1 is relieved but twice the wave-instructions are issued per visit.  Letting each lane pair walk
// two bodies at once (two loads in flight per wave) was slower still (0.43 ms).  Kept selectable.
__device__ __forceinline__ float pair_from_odd(float v) {   // quad_perm [1,1,3,3]: both lanes read the odd lane
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xF5, 0xF, 0xF, true));
}
__device__ __forceinline__ int pair_from_even(int v) {      // quad_perm [0,0,2,2]: both lanes read the even lane
    return __builtin_amdgcn_mov_dpp(v, 0xA0, 0xF, 0xF, true);
}

template <bool FAST>
__global__ __launch_bounds__(kWalkBlock) void k_bh_walk_pair(const NodeDev* __restrict__ nodes, int n_nodes,
                                                             const int* __restrict__ order, int n_order,
                                                             const float4* __restrict__ own_pos,
                                                             float4* __restrict__ acc, float g, float eps2,
                                                             float theta2, unsigned long long* __restrict__ counters,
                                                             WalkSplit split) {
    const int t = blockIdx.x * kWalkBlock + threadIdx.x;
    const int half = t & 1;
    const int tb = t >> 1;
    const int seg = blockIdx.y;
    const int s1 = split.first[seg + 1];
    const bool live = tb < n_order;
    const int b = live ? order[tb] : 0;
    const float4 p = live ? own_pos[b] : make_float4(0.f, 0.f, 0.f, 0.f);
    float ax = 0.f, ay = 0.f, az = 0.f;
    unsigned int n_acc = 0, n_vis = 0;
    int i = live ? walk_entry(nodes, split, seg, p, theta2) : s1;
    const unsigned half16 = unsigned(half) * 16u;
    const char* __restrict__ base = reinterpret_cast<const char*>(nodes);
    while (i < s1) {
        const float4 A = *reinterpret_cast<const float4*>(base + ((unsigned(i) << 5) + half16));  // even lane: {com, mass}
        asm volatile("" :: "v"(A.w));   // one whole 16-byte load (see k_bh_walk)
        const float Bx = pair_from_odd(A.x), By = pair_from_odd(A.y);                               // width^2, skip link
        const float rx = A.x - p.x, ry = A.y - p.y, rz = A.z - p.z;        // :190
        const float r2 = (rx * rx + ry * ry) + rz * rz;                     // :191
        ++n_vis;
        int next = i + 1;
        if (Bx < theta2 * r2) {                                             // :192
            float k;
            if (FAST) {
                const float rinv = __builtin_amdgcn_rsqf(r2 + eps2);
                k = (g * A.w) * ((rinv * rinv) * rinv);
                ax = __builtin_fmaf(rx, k, ax); ay = __builtin_fmaf(ry, k, ay); az = __builtin_fmaf(rz, k, az);
            } else {
                const float r_dist = __builtin_sqrtf(r2 + eps2);            // :193
                const float r_cubed = r_dist * r_dist * r_dist;             // :194
                k = ((g * A.w) / r_cubed);                                  // :195
                ax += rx * k; ay += ry * k; az += rz * k;
            }
            ++n_acc;
            next = __float_as_int(By);
        }
        i = pair_from_even(next);
    }
    if (live && !half)
        *(split.n_seg > 1 ? split.planes + size_t(seg) * split.plane_stride + tb : acc + b) =
            make_float4(ax, ay, az, split.store_work ? float(n_vis) : 0.f);  // :260
    if (half) { n_acc = 0; n_vis = 0; }
    for (int off = 32; off > 0; off >>= 1) {
        n_acc += __shfl_down(n_acc, off);
        n_vis += __shfl_down(n_vis, off);
    }
    if ((threadIdx.x & 63) == 0 && counters) {
        // one atomic pair per wave, spread over kCounterSlots address pairs: 16 384 atomics on ONE address
        // pair serialise in L2 at ~13 ns each (0.21 ms per walk at 8 segments, measured with theta2 = 1e9)
        const unsigned slot = (blockIdx.x + blockIdx.y * gridDim.x) & (kCounterSlots - 1);
        atomicAdd(&counters[2 * slot], (unsigned long long)n_acc);
        atomicAdd(&counters[2 * slot + 1], (unsigned long long)n_vis);
    }
}

#endif  // NBODY_TUNING

// Strict math, reference leaf rule: the reference's NESTED sums (barnes_hut.rs:196-202: every opened
// cell folds its children's results left to right from zero and hands the sum up).  One running sum
// per lane reproduces the visits but not that association; this kernel keeps, per lane, the open
// cells' partial sums: the innermost in registers, the others on a stack in global memory
// ([depth][lane], touched only when a cell is opened or finished).  Every add the reference performs
// is performed, in its order -- including `acc += 0` for a leaf that fails the test -- so the
// accelerations equal the oracle's bit for bit.  The node range is not split (the association would
// change); this is the parity path, the fast path is k_bh_walk.
__global__ __launch_bounds__(kWalkBlock) void k_bh_walk_nested(const NodeDev* __restrict__ nodes, int n_nodes,
                                                               const int* __restrict__ order, int n_order,
                                                               const float4* __restrict__ own_pos,
                                                               float4* __restrict__ acc, float g, float eps2,
                                                               float theta2, unsigned long long* __restrict__ counters,
                                                               float4* __restrict__ stack, size_t stack_stride,
                                                               const int* __restrict__ poison, const int* __restrict__ n_order_dev) {
    const int t = blockIdx.x * kWalkBlock + threadIdx.x;
    unsigned int n_acc = 0, n_vis = 0;
    if (poison && *poison) return;
    if (n_order_dev) n_order = min(n_order, *n_order_dev);
    if (t < n_order) {
        const int b = order[t];
        const float4 p = own_pos[b];
        float sx = 0.f, sy = 0.f, sz = 0.f;   // the innermost open cell's running sum
        int end = 0;                          // ... and the index after its subtree
        int d = 0;                            // open cells
        float ox = 0.f, oy = 0.f, oz = 0.f;   // calc_force(root)
        int i = 0;
        bool done = false;
        while (!done) {
            const float4 A = nodes[i].a;
            const float4 B = nodes[i].b;
            asm volatile("" :: "v"(A.w), "v"(B.y));
            const float rx = A.x - p.x, ry = A.y - p.y, rz = A.z - p.z;        // :190
            const float r2 = (rx * rx + ry * ry) + rz * rz;                     // :191
            const int skip = __float_as_int(B.y);
            ++n_vis;
            float fx = 0.f, fy = 0.f, fz = 0.f;
            bool value = true;               // this visit yields a value for the enclosing cell
            if (B.x < theta2 * r2) {                                            // :192
                const float r_dist = __builtin_sqrtf(r2 + eps2);                // :193
                const float r_cubed = r_dist * r_dist * r_dist;                 // :194
                const float k = ((g * A.w) / r_cubed);                          // :195
                fx = rx * k; fy = ry * k; fz = rz * k;
                ++n_acc;
                i = skip;
            } else if (skip == i + 1) {      // a leaf (or the empty root): no children, the fold of nothing is 0 (:197-202)
                i = skip;
            } else {                         // open the cell: its children fold into a fresh sum
                if (d > 0) stack[size_t(d) * stack_stride + t] = make_float4(sx, sy, sz, __int_as_float(end));
                ++d;
                sx = sy = sz = 0.f;
                end = skip;
                i = i + 1;
                value = false;
            }
            if (value) {
                if (d == 0) { ox = fx; oy = fy; oz = fz; done = true; }        // the root itself was accepted (or is a leaf)
                else { sx += fx; sy += fy; sz += fz; }                          // acc += child result
            }
            while (!done && d > 0 && i == end) {   // the innermost cell is finished: hand its sum up
                const float vx = sx, vy = sy, vz = sz;
                --d;
                if (d == 0) { ox = vx; oy = vy; oz = vz; done = true; }
                else {
                    const float4 up = stack[size_t(d) * stack_stride + t];
                    sx = up.x + vx; sy = up.y + vy; sz = up.z + vz;
                    end = __float_as_int(up.w);
                }
            }
        }
        acc[b] = make_float4(ox, oy, oz, 0.f);  // :260
    }
    for (int off = 32; off > 0; off >>= 1) {
        n_acc += __shfl_down(n_acc, off);
        n_vis += __shfl_down(n_vis, off);
    }
    if ((threadIdx.x & 63) == 0 && counters) {
        // one atomic pair per wave, spread over kCounterSlots address pairs: 16 384 atomics on ONE address
        // pair serialise in L2 at ~13 ns each (0.21 ms per walk at 8 segments, measured with theta2 = 1e9)
        const unsigned slot = (blockIdx.x + blockIdx.y * gridDim.x) & (kCounterSlots - 1);
        atomicAdd(&counters[2 * slot], (unsigned long long)n_acc);
        atomicAdd(&counters[2 * slot + 1], (unsigned long long)n_vis);
    }
}


#ifdef NBODY_TUNING   // ---- experimental walks (variants 4, 5, 3) and the stamp read-out: tuning build only
// ---- Variant 4: wave-cooperative walk over a window of node records in LDS.
// The 64 lanes of a wave (64 tree-order neighbours) step through the UNION of their node sequences with a
// wave-uniform node index; lane l takes part in node i iff i >= resume[l] (= its own walk would visit i), the wave
// goes to i + 1 while any taking-part lane opens the node, else to the node's skip link.  tools/bh_visit_hist.py:
// at N = 65 536, theta = 0.5 the union is 3 368 nodes per wave against 1 994 for the wave's longest single walk
// (1 832 on average), no step without a taking-part lane -- 1.7x the iterations of the per-lane walk, but every
// iteration reads ONE record for all lanes instead of up to 64 different ones (the per-lane walk is bound by the
// L1's tag look-ups: 1.24 cache lines per lane-visit).  k_bh_walk_wave (variant 1) fetched that record with a scalar
// load per iteration and was bound by its latency; here the wave keeps a WINDOW of kWin consecutive records in
// registers (lane l: record base + l), filled by one coalesced load (2 KB) when the index leaves it (every ~6
// iterations; the next kWin records are requested ahead), and hands the current record round with v_readlane: it
// arrives in SGPRs, no LDS, no barrier.  (A first form kept the window in LDS and read it as a broadcast: 500
// cycles per iteration on the longest waves, the LDS round trip and the exposed window fills.)  Per lane the accepted
// nodes, their order, the arithmetic and the counters are those of k_bh_walk: same bits.
constexpr int kWin = 64;

__device__ __forceinline__ float lane_value(float v, int l) {   // VGPR of lane l (wave-uniform l) -> SGPR
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// walk_entry for a whole wave at once: the ancestors of the segment's first node are the same for every lane, so
// their records are fetched by ONE gather (lane k: ancestor k) instead of a dependent chain of loads per lane
// (measured: 5.7 us per wave, a fifth of all wave-slot time at 48 segments), then every lane replays the opening tests
// on the records handed round with v_readlane.  Returns the first node >= first[seg] the lane's walk visits.
template <bool DIRECT>
__device__ __forceinline__ int walk_entry_coop(const NodeDev* __restrict__ nodes, const WalkSplit& sp, int seg,
                                               const float4 p, float theta2, int lane) {
    const int na = sp.n_anc[seg];
    int res = -1;
    for (int k0 = 0; k0 < na; k0 += 64) {
        const int cnt = min(64, na - k0);
        float4 A = make_float4(0.f, 0.f, 0.f, 0.f), B = A;
        if (lane < cnt) {
            const int j = sp.anc[seg * kMaxAnc + k0 + lane];
            A = nodes[j].a;
            B = nodes[j].b;
        }
        for (int k = 0; k < cnt; ++k) {
            const float x = lane_value(A.x, k), y = lane_value(A.y, k), z = lane_value(A.z, k);
            const float w2 = lane_value(B.x, k);
            const int skip = __builtin_amdgcn_readlane(__float_as_int(B.y), k);
            const float rx = x - p.x, ry = y - p.y, rz = z - p.z;
            const float r2 = (rx * rx + ry * ry) + rz * rz;
            const bool out = (DIRECT && r2 < 1e-10f) || (w2 < theta2 * r2);   // accepted (or skipped whole): the walk resumes behind its subtree
            if (res < 0 && out) res = skip;
        }
    }
    return res < 0 ? sp.first[seg] : res;
}

template <bool DIRECT, bool DBG = false>
__global__ __launch_bounds__(64) void k_bh_walk_coop(const NodeDev* __restrict__ nodes, int n_nodes,
                                                     const int* __restrict__ order, int n_order,
                                                     const float4* __restrict__ own_pos, float4* __restrict__ acc,
                                                     float g, float eps2, float theta2,
                                                     unsigned long long* __restrict__ counters, WalkSplit split) {
    const int lane = threadIdx.x;
    const unsigned long long r_start = DBG ? __builtin_amdgcn_s_memrealtime() : 0ull;
    int seg = blockIdx.y;
    if (split.diag_first) {   // heaviest (nearest) segments of a body group first, see k_bh_walk
        const int K = gridDim.y;
        const int diag = int((long long)blockIdx.x * K / gridDim.x);
        const int kk = blockIdx.y;
        const int off = (kk & 1) ? (kk + 1) / 2 : -(kk / 2);
        seg = ((diag + off) % K + K) % K;
    }
    const int s1 = split.first[seg + 1];
    const int t = blockIdx.x * 64 + lane;
    const bool live = t < n_order;
    const int b = live ? order[t] : 0;
    const float4 p = live ? own_pos[b] : make_float4(0.f, 0.f, 0.f, 0.f);
    float ax = 0.f, ay = 0.f, az = 0.f;
    int resume = walk_entry_coop<DIRECT>(nodes, split, seg, p, theta2, lane);
    if (!live) resume = 0x7fffffff;
    int first = resume;
    for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off));
    int i = __builtin_amdgcn_readfirstlane(first);   // wave-uniform node index: the first node any lane visits
    unsigned int n_acc = 0, n_vis = 0;               // wave totals (scalar)
    unsigned long long r_mid = 0;
    unsigned int n_it = 0, n_fill = 0;
    if (DBG) r_mid = __builtin_amdgcn_s_memrealtime();
    // the window: lane l holds record base + l in registers; the next kWin records are requested as soon as a window
    // is in place (the index only moves forward) and are usually there when the index gets to them
    int base = i;
    float4 wa = make_float4(0.f, 0.f, 0.f, 0.f), wb = wa, pa = wa, pb = wa;
    if (i < s1) {
        const int idx = min(base + lane, n_nodes - 1);
        wa = nodes[idx].a; wb = nodes[idx].b;
        const int pidx = min(base + kWin + lane, n_nodes - 1);
        pa = nodes[pidx].a; pb = nodes[pidx].b;
        if (DBG) ++n_fill;
    }
    while (i < s1) {
        if (DBG) ++n_it;
        if (unsigned(i - base) >= unsigned(kWin)) {  // the index left the window
            if (DBG) ++n_fill;
            if (unsigned(i - base - kWin) < unsigned(kWin)) {   // ... into the records requested ahead
                wa = pa; wb = pb;
                base += kWin;
            } else {                                            // ... beyond them: fetch from where it is now
                base = i;
                const int idx = min(base + lane, n_nodes - 1);
                wa = nodes[idx].a; wb = nodes[idx].b;
            }
            const int pidx = min(base + kWin + lane, n_nodes - 1);
            pa = nodes[pidx].a; pb = nodes[pidx].b;
        }
        const int r = i - base;
        const float nx = lane_value(wa.x, r), ny = lane_value(wa.y, r), nz = lane_value(wa.z, r), nm = lane_value(wa.w, r);
        const float w2 = lane_value(wb.x, r);
        const int skip = __builtin_amdgcn_readlane(__float_as_int(wb.y), r);
        const bool active = i >= resume;
        const float rx = nx - p.x, ry = ny - p.y, rz = nz - p.z;           // :190
        const float r2 = (rx * rx + ry * ry) + rz * rz;                     // :191
        bool accept, open;
        if (DIRECT) {
            const bool near = active && (r2 < 1e-10f);                      // llm :933-935: skipped whole
            accept = active && !near && ((w2 < theta2 * r2) || skip == i + 1);  // llm :938 accepted cell, :958-972 leaf
            open = active && !near && !accept;
            if (near) resume = skip;
        } else {
            accept = active && (w2 < theta2 * r2);                          // :192
            open = active && !accept;
        }
        if (accept) {
            const float rinv = __builtin_amdgcn_rsqf(r2 + eps2);
            const float k = (g * nm) * ((rinv * rinv) * rinv);
            ax += rx * k; ay += ry * k; az += rz * k;
            resume = skip;
        }
        n_vis += unsigned(__popcll(__builtin_amdgcn_ballot_w64(active)));
        n_acc += unsigned(__popcll(__builtin_amdgcn_ballot_w64(accept)));
        i = __builtin_amdgcn_ballot_w64(open) != 0ull ? i + 1 : skip;
    }
    if (live) *(split.n_seg > 1 ? split.planes + size_t(seg) * split.plane_stride + t : acc + b) = make_float4(ax, ay, az, 0.f);  // :260
    if (DBG) {   // per wave: start, end (100 MHz ticks), iterations | window fills << 24 | entry replay ticks << 44
        const unsigned w = blockIdx.y * gridDim.x + blockIdx.x;
        if (lane == 0 && w < 65536) {
            nbody_bh_stamps[3 * w] = r_start; nbody_bh_stamps[3 * w + 1] = __builtin_amdgcn_s_memrealtime();
            nbody_bh_stamps[3 * w + 2] = (unsigned long long)n_it | ((unsigned long long)n_fill << 24) | ((unsigned long long)min(0xFFFFull, r_mid - r_start) << 44);
        }
    }
    if (lane == 0 && counters) {
        const unsigned slot = (blockIdx.x + blockIdx.y * gridDim.x) & (kCounterSlots - 1);
        atomicAdd(&counters[2 * slot], (unsigned long long)n_acc);
        atomicAdd(&counters[2 * slot + 1], (unsigned long long)n_vis);
    }
}

// ---- Variant 5: cooperative BLOCK walk over a level-order copy of the tree (kernels_tree.hip build_bfs_layout).
// Variant 4 showed what a wave-cooperative walk is worth and what limits it: the union of 64 neighbours' walks is
// only 1.7x one body's walk, but in the pre-order array consecutive visited nodes come in runs of ~1.5, so nearly every
// iteration needs a record from a new place (a window of 64 records is refilled every 5.8 iterations and 91 % of what
// it fetches is never looked at: 2.7 GB of L2 traffic per launch, 400-500 cycles per iteration).  What IS contiguous
// is a node's list of children when the nodes are stored level by level.  So: a wave keeps a stack of
// {child block, mask of the lanes that opened the parent}; it pops a block, fetches its <= 8 records with one load
// (lanes 0..7), tests every child for the lanes of the mask, adds the accepted ones, and pushes the blocks of the
// children some lane opened.  tools/bh_visit_hist.py, N = 65 536, theta = 0.5: 606 block fetches per wave instead of
// 3 368 dependent record fetches, 5.55 children per block, at most 54 stack entries.
// Every lane evaluates exactly the opening tests of its own walk (so the counters equal k_bh_walk's and the
// oracle's); it adds the accepted monopoles of a block's children before those of their subtrees, in reverse
// orthant order -- a fixed order, but not the pre-order of k_bh_walk: accelerations agree to rounding, not bit for bit.
// The node-range split works on the records' pre-order indices: a subtree entirely outside [s0, s1) is not touched,
// a node before s0 whose subtree reaches into the range is only tested (it belongs to an earlier segment).
constexpr int kBlockStack = 320;   // entries; a walk holds <= 7 per level + 1, the device build stops at 42 levels

template <bool DIRECT, bool DBG = false>
__global__ __launch_bounds__(64) void k_bh_walk_block(const NodeDev* __restrict__ bfs, int n_nodes,
                                                      const int* __restrict__ order, int n_order,
                                                      const float4* __restrict__ own_pos, float4* __restrict__ acc,
                                                      float g, float eps2, float theta2,
                                                      unsigned long long* __restrict__ counters, WalkSplit split) {
    __shared__ uint4 stack[kBlockStack];   // {block position, mask lo, mask hi, -}
    const int lane = threadIdx.x;
    const unsigned long long r_start = DBG ? __builtin_amdgcn_s_memrealtime() : 0ull;
    int seg = blockIdx.y;
    if (split.diag_first) {   // heaviest (nearest) segments of a body group first, see k_bh_walk
        const int K = gridDim.y;
        const int diag = int((long long)blockIdx.x * K / gridDim.x);
        const int kk = blockIdx.y;
        const int off = (kk & 1) ? (kk + 1) / 2 : -(kk / 2);
        seg = ((diag + off) % K + K) % K;
    }
    const int s0 = split.first[seg], s1 = split.first[seg + 1];
    const int t = blockIdx.x * 64 + lane;
    const bool live = t < n_order;
    const int b = live ? order[t] : 0;
    const float4 p = live ? own_pos[b] : make_float4(0.f, 0.f, 0.f, 0.f);
    float ax = 0.f, ay = 0.f, az = 0.f;
    unsigned int n_acc = 0, n_vis = 0, n_blk = 0, n_it = 0;
    int sp = 0;
    {
        const unsigned long long m0 = __builtin_amdgcn_ballot_w64(live);
        if (lane == 0) stack[0] = make_uint4(0u, unsigned(m0), unsigned(m0 >> 32), 0u);   // the root: a block of one
        sp = 1;
    }
    while (sp > 0) {
        --sp;
        __syncthreads();                                   // (one wave per workgroup: orders the LDS accesses)
        const uint4 e = stack[sp];
        const int base = __builtin_amdgcn_readfirstlane(int(e.x));
        const unsigned long long M = ((unsigned long long)__builtin_amdgcn_readfirstlane(int(e.y)) & 0xFFFFFFFFull)
                                   | ((unsigned long long)__builtin_amdgcn_readfirstlane(int(e.z)) << 32);
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = make_float4(0.f, 0.f, 0.f, __int_as_float(int(0x80000000u)));
        if (lane < 8) {
            const int idx = min(base + lane, n_nodes - 1);
            ra = bfs[idx].a; rb = bfs[idx].b;
        }
        // children in the block: up to and including the first record with the last-sibling flag
        const unsigned long long lastm = __builtin_amdgcn_ballot_w64(lane < 8 && __float_as_int(rb.w) < 0);
        const int nchild = __builtin_ctzll(lastm) + 1;
        const bool active = (M >> lane) & 1ull;
        if (DBG) ++n_blk;
        for (int c = nchild - 1; c >= 0; --c) {            // reverse, so that the pushes pop in orthant order
            const int pre = __builtin_amdgcn_readlane(__float_as_int(rb.y), c);
            const int skp = __builtin_amdgcn_readlane(__float_as_int(rb.z), c);
            if (skp <= s0 || pre >= s1) continue;          // the subtree lies outside this segment
            const bool in_seg = pre >= s0;                 // else: an ancestor of the segment's first node, tested only
            if (DBG) ++n_it;
            const float nx = lane_value(ra.x, c), ny = lane_value(ra.y, c), nz = lane_value(ra.z, c), nm = lane_value(ra.w, c);
            const float w2 = lane_value(rb.x, c);
            const float rx = nx - p.x, ry = ny - p.y, rz = nz - p.z;       // :190
            const float r2 = (rx * rx + ry * ry) + rz * rz;                 // :191
            const bool leaf = skp == pre + 1;
            bool accept, open;
            if (DIRECT) {
                const bool near = active && (r2 < 1e-10f);                  // llm :933-935: skipped whole
                accept = active && !near && ((w2 < theta2 * r2) || leaf);   // llm :938 accepted cell, :958-972 leaf
                open = active && !near && !accept;
            } else {
                accept = active && (w2 < theta2 * r2);                      // :192
                open = active && !accept;
            }
            if (in_seg) {
                if (accept) {
                    const float rinv = __builtin_amdgcn_rsqf(r2 + eps2);
                    const float k = (g * nm) * ((rinv * rinv) * rinv);
                    ax += rx * k; ay += ry * k; az += rz * k;
                }
                n_vis += unsigned(__popcll(__builtin_amdgcn_ballot_w64(active)));
                n_acc += unsigned(__popcll(__builtin_amdgcn_ballot_w64(accept)));
            }
            const unsigned long long mo = __builtin_amdgcn_ballot_w64(open);
            if (mo != 0ull && !leaf && sp < kBlockStack) {
                const int child = __builtin_amdgcn_readlane(__float_as_int(rb.w), c) & 0x7fffffff;
                if (lane == 0) stack[sp] = make_uint4(unsigned(child), unsigned(mo), unsigned(mo >> 32), 0u);
                ++sp;
            }
        }
    }
    if (live) *(split.n_seg > 1 ? split.planes + size_t(seg) * split.plane_stride + t : acc + b) = make_float4(ax, ay, az, 0.f);  // :260
    if (DBG) {   // per wave: start, end (100 MHz ticks), children tested | blocks fetched << 24
        const unsigned w = blockIdx.y * gridDim.x + blockIdx.x;
        if (lane == 0 && w < 65536) {
            nbody_bh_stamps[3 * w] = r_start; nbody_bh_stamps[3 * w + 1] = __builtin_amdgcn_s_memrealtime();
            nbody_bh_stamps[3 * w + 2] = (unsigned long long)n_it | ((unsigned long long)n_blk << 24);
        }
    }
    if (lane == 0 && counters) {
        const unsigned slot = (blockIdx.x + blockIdx.y * gridDim.x) & (kCounterSlots - 1);
        atomicAdd(&counters[2 * slot], (unsigned long long)n_acc);
        atomicAdd(&counters[2 * slot + 1], (unsigned long long)n_vis);
    }
}

// ---- Variant 3: the most-visited node records staged in LDS (north_star's "cell list staged in LDS").
// tools/bh_visit_hist.py: at N = 65 536, theta = 0.5 the walks evaluate 1.2e8 opening tests on 97 179 nodes, and
// the ~2 500 nodes whose grandparent cell holds >= 1 024 bodies take 70 % of them.  tools/microbench_gather.hip:
// a CU sustains 0.675 dependent 2 x 16-byte gathers per cycle from L1/L2 with divergent lanes (k_bh_walk runs
// at ~0.6), 1.03 from LDS, and 1.2-1.3 when half to two thirds of them go to LDS and the rest to L1 -- the two
// pipes work side by side.
//
// Staged nodes are not a prefix of the pre-order array, so the walk array carries explicit links:
//   record = {com, mass | w^2, link taken when the node is accepted (its skip), link taken when it is opened
//             (its first child; for a leaf the same as the skip link), the node's pre-order index}
//   link < M: slot of the LDS table; otherwise pre-order index + M in the walk array.
// The pre-order index only decides where a segment of the node-range split ends (see WalkSplit); a sentinel
// record behind the last node ends the last one.  The visits, their order and therefore the counters and
// the per-segment sums are those of k_bh_walk; only where a record is read from differs.
__global__ __launch_bounds__(256) void k_walk_slots(const NodeDev* __restrict__ nodes, int n_nodes, int threshold, int M,
                                                    int* __restrict__ unified, int* __restrict__ info) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i > n_nodes) return;
    int u = i + M;
    if (i < n_nodes && __float_as_int(nodes[i].b.z) >= threshold) {
        const int s = atomicAdd(info, 1);   // which of the flagged nodes get a slot when there are more than M is
        if (s < M) u = s;                   // left to chance: the results do not depend on where a record lives
    }
    unified[i] = u;
}

__global__ __launch_bounds__(256) void k_walk_links(const NodeDev* __restrict__ nodes, int n_nodes, int M,
                                                    const int* __restrict__ unified, NodeDev* __restrict__ walk,
                                                    NodeDev* __restrict__ hot, int* __restrict__ info) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { info[1] = info[0]; info[0] = 0; }   // flagged count for the host's threshold control; counter reset for the next pass
    if (i > n_nodes) return;
    NodeDev r;
    if (i == n_nodes) {   // sentinel: pre-order index n_nodes ends every segment
        const float self = __int_as_float(n_nodes + M);
        r.a = make_float4(0.f, 0.f, 0.f, 0.f);
        r.b = make_float4(0.f, self, self, __int_as_float(n_nodes));
    } else {
        r = nodes[i];
        const int skip = __float_as_int(r.b.y);
        r.b = make_float4(r.b.x, __int_as_float(unified[skip]), __int_as_float(unified[i + 1]), __int_as_float(i));
    }
    walk[i] = r;
    const int u = unified[i];
    if (u < M) hot[u] = r;
}

template <bool DIRECT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_bh_walk_lds(const NodeDev* __restrict__ nodes, const NodeDev* __restrict__ walk,
                                                       const NodeDev* __restrict__ hot, int M,
                                                       const int* __restrict__ unified, const int* __restrict__ order,
                                                       int n_order, const float4* __restrict__ own_pos,
                                                       float4* __restrict__ acc, float g, float eps2, float theta2,
                                                       unsigned long long* __restrict__ counters, WalkSplit split) {
    extern __shared__ float4 lds[];   // [2 * M]: the staged records
    for (int k = threadIdx.x; k < 2 * M; k += BLOCK) lds[k] = reinterpret_cast<const float4*>(hot)[k];
    __syncthreads();
    // a workgroup's waves: consecutive (body group, segment) pairs, segment fastest -- a body group's K segments
    // sit on one CU (equal work per CU, as in k_bh_walk's grid)
    const int wv = blockIdx.x * (BLOCK / 64) + int(threadIdx.x >> 6);
    const int grp = wv / split.n_seg, seg = wv - grp * split.n_seg;
    const int t = grp * 64 + int(threadIdx.x & 63);
    unsigned int n_acc = 0, n_vis = 0;
    if (t < n_order) {
        const int s1 = split.first[seg + 1];
        const int b = order[t];
        const float4 p = own_pos[b];
        float ax = 0.f, ay = 0.f, az = 0.f;
        int u = unified[walk_entry<DIRECT>(nodes, split, seg, p, theta2)];
        const NodeDev* __restrict__ wbase = walk - M;   // indexable by link value
        while (true) {
            float4 A, B;
            if (u < M) { A = lds[2 * u]; B = lds[2 * u + 1]; }
            else { A = wbase[u].a; B = wbase[u].b; }
            asm volatile("" :: "v"(A.w), "v"(B.y), "v"(B.z));   // both 16-byte loads whole, ahead of the branches (see k_bh_walk)
            if (__float_as_int(B.w) >= s1) break;
            const float rx = A.x - p.x, ry = A.y - p.y, rz = A.z - p.z;        // :190
            const float r2 = (rx * rx + ry * ry) + rz * rz;                     // :191
            ++n_vis;
            const int l_skip = __float_as_int(B.y), l_open = __float_as_int(B.z);
            if (DIRECT) {
                if (r2 < 1e-10f) { u = l_skip; continue; }                      // llm :933-935
                if (B.x < theta2 * r2 || l_open == l_skip) {                    // llm :938 accepted cell, :958-972 leaf
                    const float rinv = __builtin_amdgcn_rsqf(r2 + eps2);
                    const float k = (g * A.w) * ((rinv * rinv) * rinv);
                    ax += rx * k; ay += ry * k; az += rz * k;
                    ++n_acc;
                    u = l_skip;
                } else {
                    u = l_open;
                }
                continue;
            }
            if (B.x < theta2 * r2) {                                            // :192
                const float rinv = __builtin_amdgcn_rsqf(r2 + eps2);
                const float k = (g * A.w) * ((rinv * rinv) * rinv);
                ax += rx * k; ay += ry * k; az += rz * k;
                ++n_acc;
                u = l_skip;
            } else {
                u = l_open;
            }
        }
        *(split.n_seg > 1 ? split.planes + size_t(seg) * split.plane_stride + t : acc + b) =
            make_float4(ax, ay, az, split.store_work ? float(n_vis) : 0.f);  // :260
    }
    for (int off = 32; off > 0; off >>= 1) {
        n_acc += __shfl_down(n_acc, off);
        n_vis += __shfl_down(n_vis, off);
    }
    if ((threadIdx.x & 63) == 0 && counters) {
        const unsigned slot = unsigned(wv) & (kCounterSlots - 1);
        atomicAdd(&counters[2 * slot], (unsigned long long)n_acc);
        atomicAdd(&counters[2 * slot + 1], (unsigned long long)n_vis);
    }
}

}  // namespace nbody
extern "C" int nbody_bh_read_stamps(unsigned long long* out, int n_waves) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(nbody::nbody_bh_stamps), sizeof(unsigned long long) * 3 * n_waves) == hipSuccess ? 0 : -1;
}
namespace nbody {
#endif  // NBODY_TUNING

// KICK: integrate_after_force (shared.rs:141-148) rides along, as in k_bf_sym_reduce
template <bool KICK>
__global__ __launch_bounds__(256) void k_bh_reduce(const float4* __restrict__ planes, int n_seg, size_t plane_stride,
                                                   const int* __restrict__ order, int n_order,
                                                   float4* __restrict__ acc, float4* __restrict__ pos,
                                                   float4* __restrict__ vel, float dt, int* __restrict__ poison,
                                                   const int* __restrict__ n_order_dev, int store_work) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (poison && *poison) return;
    if (KICK && poison && t == 0) atomicAdd(poison + 1, 1);   // a step of an unsynchronised run is complete
    if (n_order_dev) n_order = min(n_order, *n_order_dev);
    if (t >= n_order) return;
    const int b = order[t];
    float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
    for (int k = 0; k < n_seg; ++k) {  // segment order = the order the single walk adds them in
        const float4 v = planes[size_t(k) * plane_stride + t];   // (by tree-order position: coalesced; by body it was a gather, 18 us)
        sx += v.x; sy += v.y; sz += v.z; sw += v.w;
    }
    acc[b] = make_float4(sx, sy, sz, store_work ? sw : 0.f);   // (w: the body's visit count, a sum of small integers: exact)
    if (KICK) {
        float4 p = pos[b], v = vel[b];
        v.x += sx * dt;                 // shared.rs:144
        v.y += sy * dt;
        v.z += sz * dt;
        p.x += (v.x * 0.5f) * dt;       // shared.rs:146
        p.y += (v.y * 0.5f) * dt;
        p.z += (v.z * 0.5f) * dt;
        vel[b] = v;
        pos[b] = p;
    }
}

// The same with Q waves per 64 bodies: wave q adds the planes of the segments [q n_seg / Q, (q + 1) n_seg / Q) and wave 0
// adds the Q partial sums -- another association than the plain form's, fixed all the same (deterministic).  One thread
// per body is 256 workgroups at N = 65 536, one per CU, each walking 16 planes in a row: 11 us of mostly latency.
template <bool KICK, int Q>
__global__ __launch_bounds__(64 * Q) void k_bh_reduce_split(const float4* __restrict__ planes, int n_seg, size_t plane_stride,
                                                            const int* __restrict__ order, int n_order, float4* __restrict__ acc,
                                                            float4* __restrict__ pos, float4* __restrict__ vel, float dt,
                                                            int* __restrict__ poison, const int* __restrict__ n_order_dev, int store_work) {
    __shared__ float part[Q][4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    if (poison && *poison) return;
    if (KICK && poison && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(poison + 1, 1);   // a step of an unsynchronised run is complete
    if (n_order_dev) n_order = min(n_order, *n_order_dev);
    float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
    if (t < n_order) {
        const int k0 = n_seg * q / Q, k1 = n_seg * (q + 1) / Q;
        for (int k = k0; k < k1; ++k) {
            const float4 v = planes[size_t(k) * plane_stride + t];
            sx += v.x; sy += v.y; sz += v.z; sw += v.w;
        }
    }
    part[q][0][lane] = sx; part[q][1][lane] = sy; part[q][2][lane] = sz; part[q][3][lane] = sw;
    __syncthreads();
    if (q != 0 || t >= n_order) return;
    sx = part[0][0][lane]; sy = part[0][1][lane]; sz = part[0][2][lane]; sw = part[0][3][lane];
    for (int w = 1; w < Q; ++w) { sx += part[w][0][lane]; sy += part[w][1][lane]; sz += part[w][2][lane]; sw += part[w][3][lane]; }
    const int b = order[t];
    acc[b] = make_float4(sx, sy, sz, store_work ? sw : 0.f);
    if (KICK) {
        float4 p = pos[b], v = vel[b];
        v.x += sx * dt; v.y += sy * dt; v.z += sz * dt;                                       // shared.rs:144
        p.x += (v.x * 0.5f) * dt; p.y += (v.y * 0.5f) * dt; p.z += (v.z * 0.5f) * dt;         // shared.rs:146
        vel[b] = v;
        pos[b] = p;
    }
}

#ifdef NBODY_TUNING
// The experimental walks of the tuning build (Tuning::bh_walk_variant 1..5) and the stamped instantiations
// (Tuning::bh_walk_debug); returns false when the default walk is to run.
static bool launch_walk_variant(hipStream_t s, const Shard& sh, const TreeDev& t, float g, float g_soft2, float theta2, int fast_math,
                                unsigned long long* counters, int leaf_direct, const WalkSplit& sp, dim3 grid) {
    if (fast_math && tuning().bh_walk_variant == 3 && t.hot_cap > 0 && t.walk) {
        const int M = t.hot_cap;
        const dim3 pg((t.n_nodes + 1 + 255) / 256);
        hipLaunchKernelGGL(k_walk_slots, pg, dim3(256), 0, s, reinterpret_cast<const NodeDev*>(t.nodes), t.n_nodes, t.hot_threshold, M, t.unified, t.hot_info);
        hipLaunchKernelGGL(k_walk_links, pg, dim3(256), 0, s, reinterpret_cast<const NodeDev*>(t.nodes), t.n_nodes, M, t.unified,
                           reinterpret_cast<NodeDev*>(t.walk), reinterpret_cast<NodeDev*>(t.hot), t.hot_info);
        const int groups = (t.n_order + 63) / 64;
        const long long waves = (long long)groups * t.n_split;
        const size_t lds_bytes = size_t(M) * sizeof(NodeDev);
#define WALK_LDS(DIRECT, BLK) do {                                                                                          \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bh_walk_lds<DIRECT, BLK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            hipLaunchKernelGGL((k_bh_walk_lds<DIRECT, BLK>), dim3(unsigned((waves + BLK / 64 - 1) / (BLK / 64))), dim3(BLK), lds_bytes, s,   \
                               reinterpret_cast<const NodeDev*>(t.nodes), reinterpret_cast<const NodeDev*>(t.walk),         \
                               reinterpret_cast<const NodeDev*>(t.hot), M, t.unified, t.order, t.n_order, sh.own_pos(), sh.acc, g, g_soft2, theta2, counters, sp); } while (0)
        if (tuning().bh_walk_lds_block == 512) { if (leaf_direct) WALK_LDS(true, 512); else WALK_LDS(false, 512); }
        else if (tuning().bh_walk_lds_block == 256) { if (leaf_direct) WALK_LDS(true, 256); else WALK_LDS(false, 256); }
        else { if (leaf_direct) WALK_LDS(true, 1024); else WALK_LDS(false, 1024); }
#undef WALK_LDS
        return true;
    }
    if (fast_math && tuning().bh_walk_variant == 5 && t.bfs && t.n_nodes > 0) {
        (void)build_bfs_layout(s, t.nodes, t.n_nodes, t.bfs_ws, t.bfs_cap, t.bfs);
#define WALK_BLOCK(...) hipLaunchKernelGGL((k_bh_walk_block<__VA_ARGS__>), grid, dim3(64), 0, s, reinterpret_cast<const NodeDev*>(t.bfs), t.n_nodes, t.order, t.n_order, sh.own_pos(), sh.acc, g, g_soft2, theta2, counters, sp)
        if (tuning().bh_walk_debug && !leaf_direct) WALK_BLOCK(false, true);
        else if (leaf_direct) WALK_BLOCK(true);
        else WALK_BLOCK(false);
#undef WALK_BLOCK
        return true;
    }
    if (fast_math && tuning().bh_walk_variant == 4) {
        if (tuning().bh_walk_debug && !leaf_direct) hipLaunchKernelGGL((k_bh_walk_coop<false, true>), grid, dim3(64), 0, s, reinterpret_cast<const NodeDev*>(t.nodes), t.n_nodes, t.order, t.n_order, sh.own_pos(), sh.acc, g, g_soft2, theta2, counters, sp);
        else if (leaf_direct) hipLaunchKernelGGL((k_bh_walk_coop<true>), grid, dim3(64), 0, s, reinterpret_cast<const NodeDev*>(t.nodes), t.n_nodes, t.order, t.n_order, sh.own_pos(), sh.acc, g, g_soft2, theta2, counters, sp);
        else hipLaunchKernelGGL((k_bh_walk_coop<false>), grid, dim3(64), 0, s, reinterpret_cast<const NodeDev*>(t.nodes), t.n_nodes, t.order, t.n_order, sh.own_pos(), sh.acc, g, g_soft2, theta2, counters, sp);
        return true;
    }
    // the alternative walks 1 and 2 are fast-math experiments with the reference leaf rule only
    const int variant = (leaf_direct || !fast_math) ? 0 : (tuning().bh_walk_variant >= 3 ? 0 : tuning().bh_walk_variant);
    if (variant == 2) grid.x = (2 * t.n_order + kWalkBlock - 1) / kWalkBlock;
#define WALK(K, ...) hipLaunchKernelGGL((K<__VA_ARGS__>), grid, dim3(kWalkBlock), 0, s, reinterpret_cast<const NodeDev*>(t.nodes), t.n_nodes, t.order, t.n_order, sh.own_pos(), sh.acc, g, g_soft2, theta2, counters, sp)
    if (variant == 1) { WALK(k_bh_walk_wave, true); return true; }
    if (variant == 2) { WALK(k_bh_walk_pair, true); return true; }
    if (tuning().bh_walk_debug && fast_math && !leaf_direct) { WALK(k_bh_walk, true, false, true); return true; }   // per-wave stamps (tools/bh_wave_times.py)
#undef WALK
    return false;
}
#endif  // NBODY_TUNING

// Split points of ANY pre-order node array with skip links (spatial shards: the array of held nodes has no build arrays to
// take them from): segment k starts at node range[0] + (range[1] - range[0]) k / n_split; its ancestors are found by
// walking down from the root -- a node whose skip link lies beyond t holds t in its subtree.  One thread per split point.
__global__ void k_walk_split_scan(const NodeDev* __restrict__ nodes, const int* __restrict__ range, int n_split, int* __restrict__ first,
                                  int* __restrict__ n_anc, int* __restrict__ anc, int first_given) {
    const int s = threadIdx.x;
    if (s >= n_split) return;
    const int b = range[0], e = range[1];
    const int t = first_given ? first[s] : b + int((long long)(e - b) * s / n_split);
    if (!first_given) {
        first[s] = t;
        if (s == n_split - 1) first[n_split] = e;
    }
    int i = b, na = 0;
    while (i < t && na < kMaxAnc) {
        const int skip = __float_as_int(nodes[i].b.y);
        if (skip > t) { anc[s * kMaxAnc + na++] = i; i = i + 1; }
        else i = skip;
    }
    n_anc[s] = na;
}
// first_given: first[0 .. n_split] are the caller's (spatial shards cut at GLOBAL node indices, so that what a rank holds
// beyond the nodes its bodies visit does not move the cuts); otherwise equal parts of the array
void launch_walk_split_scan(hipStream_t s, const float4* nodes, const int* range, int n_split, int* first, int* n_anc, int* anc, int first_given) {
    hipLaunchKernelGGL(k_walk_split_scan, dim3(1), dim3(64 * ((n_split + 63) / 64)), 0, s, reinterpret_cast<const NodeDev*>(nodes), range, n_split,
                       first, n_anc, anc, first_given);
}

void launch_bh_walk(hipStream_t s, const Shard& sh, const TreeDev& t, float g, float g_soft2, float theta2,
                    int fast_math, unsigned long long* counters, int leaf_direct, const float* kick_dt, int* kicked) {
    if (kicked) *kicked = 0;
    if (t.n_order <= 0) return;
    if (t.nested_stack && !fast_math && !leaf_direct) {  // strict math: always the parity kernel
        hipLaunchKernelGGL(k_bh_walk_nested, dim3((t.n_order + kWalkBlock - 1) / kWalkBlock), dim3(kWalkBlock), 0, s,
                           reinterpret_cast<const NodeDev*>(t.nodes), t.n_nodes, t.order, t.n_order, sh.own_pos(), sh.acc, g,
                           g_soft2, theta2, counters, t.nested_stack, t.nested_stride, t.poison, t.n_order_dev);
        return;
    }
    WalkSplit sp;
    sp.n_seg = t.n_split; sp.first = t.split_first; sp.anc = t.split_anc; sp.n_anc = t.split_n_anc;
    sp.planes = t.split_planes; sp.plane_stride = t.split_stride;
    sp.diag_first = tuning().bh_walk_order;
    sp.poison = t.poison; sp.n_order_dev = t.n_order_dev;
    sp.store_work = t.store_work;
    sp.xcd_blocks = 0;
    dim3 grid((t.n_order + kWalkBlock - 1) / kWalkBlock, t.n_split);
    bool launched = false;
#ifdef NBODY_TUNING   // the experimental walks (fast math only; 1 and 2: reference leaf rule only) and the stamped instantiations
    launched = launch_walk_variant(s, sh, t, g, g_soft2, theta2, fast_math, counters, leaf_direct, sp, grid);
#endif
    const int bpl = walk_plan(size_t(t.n_order), fast_math != 0, 1 << 20, theta2).bodies_per_lane;
    if (!launched && bpl >= 2) {   // several bodies per lane (k_bh_walk_duo)
        const int groups = (t.n_order + bpl - 1) / bpl;
        const bool by_xcd = tuning().bh_walk_xcd != 0;
#define DUO(BLK, BPL, ...) do { const int gx = (groups + BLK - 1) / BLK, gx8 = (gx + 7) / 8 * 8; sp.xcd_blocks = by_xcd ? gx8 / 8 : 0; hipLaunchKernelGGL((k_bh_walk_duo<__VA_ARGS__, BLK, BPL>), dim3(by_xcd ? gx8 : gx, t.n_split), dim3(BLK), 0, s, reinterpret_cast<const NodeDev*>(t.nodes), t.n_nodes, t.order, t.n_order, sh.own_pos(), sh.acc, g, g_soft2, theta2, counters, sp); } while (0)
#define DUO_B(BLK, BPL) do { if (leaf_direct) { if (fast_math) DUO(BLK, BPL, true, true); else DUO(BLK, BPL, false, true); } else { if (fast_math) DUO(BLK, BPL, true, false); else DUO(BLK, BPL, false, false); } } while (0)
#define DUO_A(BLK) do { if (bpl == 8) DUO_B(BLK, 8); else if (bpl == 6) DUO_B(BLK, 6); else if (bpl == 4) DUO_B(BLK, 4); else if (bpl == 3) DUO_B(BLK, 3); else DUO_B(BLK, 2); } while (0)
        if (t.n_split <= 2) DUO_A(256); else DUO_A(64);
#undef DUO_A
#undef DUO_B
#undef DUO
        launched = true;
    }
    if (!launched) {
#define WALK(K, ...) hipLaunchKernelGGL((K<__VA_ARGS__>), grid, dim3(kWalkBlock), 0, s, reinterpret_cast<const NodeDev*>(t.nodes), t.n_nodes, t.order, t.n_order, sh.own_pos(), sh.acc, g, g_soft2, theta2, counters, sp)
        if (t.n_split <= 2) {
            // enough bodies to fill the chip with one or two segments (N >= ~2.5e5): 256-thread workgroups are up to
            // 10 % faster there (6.2 against 6.8 ms at N = 2^20); with more segments one-wave workgroups win (kWalkBlock)
            grid = dim3((t.n_order + 255) / 256, t.n_split);
#define WALK256(...) hipLaunchKernelGGL((k_bh_walk<__VA_ARGS__, false, 256>), grid, dim3(256), 0, s, reinterpret_cast<const NodeDev*>(t.nodes), t.n_nodes, t.order, t.n_order, sh.own_pos(), sh.acc, g, g_soft2, theta2, counters, sp)
            if (leaf_direct) { if (fast_math) WALK256(true, true); else WALK256(false, true); }
            else { if (fast_math) WALK256(true, false); else WALK256(false, false); }
#undef WALK256
        }
        else if (leaf_direct) { if (fast_math) WALK(k_bh_walk, true, true); else WALK(k_bh_walk, false, true); }
        else { if (fast_math) WALK(k_bh_walk, true); else WALK(k_bh_walk, false); }
#undef WALK
    }
    if (t.n_split >= 8 && fast_math && tuning().bh_reduce_split) {   // (strict math never splits; the plain form keeps the single walk's order)
        const dim3 rg((t.n_order + 63) / 64);
        if (kick_dt) {
            hipLaunchKernelGGL((k_bh_reduce_split<true, 4>), rg, dim3(256), 0, s, t.split_planes, t.n_split, t.split_stride, t.order, t.n_order,
                               sh.acc, sh.own_pos(), sh.vel, *kick_dt, sh.poison, t.n_order_dev, t.store_work);
            if (kicked) *kicked = 1;
        } else {
            hipLaunchKernelGGL((k_bh_reduce_split<false, 4>), rg, dim3(256), 0, s, t.split_planes, t.n_split, t.split_stride, t.order, t.n_order,
                               sh.acc, sh.own_pos(), sh.vel, 0.f, sh.poison, t.n_order_dev, t.store_work);
        }
    } else if (t.n_split > 1) {
        const dim3 rg((t.n_order + 255) / 256);
        if (kick_dt) {
            hipLaunchKernelGGL(k_bh_reduce<true>, rg, dim3(256), 0, s, t.split_planes, t.n_split, t.split_stride, t.order,
                               t.n_order, sh.acc, sh.own_pos(), sh.vel, *kick_dt, sh.poison, t.n_order_dev, t.store_work);
            if (kicked) *kicked = 1;
        } else {
            hipLaunchKernelGGL(k_bh_reduce<false>, rg, dim3(256), 0, s, t.split_planes, t.n_split, t.split_stride, t.order,
                               t.n_order, sh.acc, sh.own_pos(), sh.vel, 0.f, sh.poison, t.n_order_dev, t.store_work);
        }
    }
}

}  // namespace nbody
