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
// accepted-node and visited-node counts equal the reference's.  Accepted monopoles are added
// into one running sum; the reference nests the sums per tree level, which differs by rounding
// only (tolerance in tests/test_bh_gpu.py).
//
// Bodies are walked in tree (depth-first leaf) order, so the 64 lanes of a wave hold spatial
// neighbours and follow nearly the same path: their node reads coalesce into a few 32-byte
// sectors that stay in L1/L2 (the whole node array is 32 B x n_nodes, ~4 MB at N = 65 536).
#include "kernels.h"

namespace nbody {

constexpr int kWalkBlock = 256;

template <bool FAST>
__global__ __launch_bounds__(kWalkBlock) void k_bh_walk(const float4* __restrict__ node_a,
                                                        const float4* __restrict__ node_b, int n_nodes,
                                                        const int* __restrict__ order, int n_order,
                                                        const float4* __restrict__ own_pos, float4* __restrict__ acc,
                                                        float g, float eps2, float theta2,
                                                        unsigned long long* __restrict__ counters) {
    const int t = blockIdx.x * kWalkBlock + threadIdx.x;
    unsigned int n_acc = 0, n_vis = 0;
    if (t < n_order) {
        const int b = order[t];
        const float4 p = own_pos[b];
        float ax = 0.f, ay = 0.f, az = 0.f;
        int i = 0;
        while (i < n_nodes) {
            const float4 A = node_a[i];
            const float4 B = node_b[i];
            const float rx = A.x - p.x, ry = A.y - p.y, rz = A.z - p.z;        // :190
            const float r2 = (rx * rx + ry * ry) + rz * rz;                     // :191
            ++n_vis;
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
        acc[b] = make_float4(ax, ay, az, 0.f);                                  // overwrite, :260
    }
    // one atomic pair per wave
    for (int off = 32; off > 0; off >>= 1) {
        n_acc += __shfl_down(n_acc, off);
        n_vis += __shfl_down(n_vis, off);
    }
    if ((threadIdx.x & 63) == 0 && counters) {
        atomicAdd(&counters[0], (unsigned long long)n_acc);
        atomicAdd(&counters[1], (unsigned long long)n_vis);
    }
}

void launch_bh_walk(hipStream_t s, const Shard& sh, const TreeDev& t, float g, float g_soft2, float theta2,
                    int fast_math, unsigned long long* counters) {
    if (t.n_order <= 0) return;
    int blocks = (t.n_order + kWalkBlock - 1) / kWalkBlock;
    if (fast_math)
        hipLaunchKernelGGL(k_bh_walk<true>, dim3(blocks), dim3(kWalkBlock), 0, s, t.node_a, t.node_b, t.n_nodes,
                           t.order, t.n_order, sh.own_pos(), sh.acc, g, g_soft2, theta2, counters);
    else
        hipLaunchKernelGGL(k_bh_walk<false>, dim3(blocks), dim3(kWalkBlock), 0, s, t.node_a, t.node_b, t.n_nodes,
                           t.order, t.n_order, sh.own_pos(), sh.acc, g, g_soft2, theta2, counters);
}

}  // namespace nbody
