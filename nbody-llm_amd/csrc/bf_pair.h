// bf_pair.h -- device helpers shared by the register-resident all-pairs kernels
// (kernels_bf_sym.hip, kernels_bf_cross.hip): the lane rotate through the LDS crossbar and the
// stage-ordered evaluation of IPT pairs per lane.
#pragma once
#include <hip/hip_runtime.h>

namespace nbody {
namespace {

// rotate by one lane through the LDS crossbar: lane l receives lane (l-1)&63's value.  No LDS memory
// is touched and no VALU slot is spent; the result arrives ~60+ cycles later, behind other work.
__device__ __forceinline__ float rotl(float v, int src_lane_x4) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane_x4, __float_as_int(v)));
}

#define PAD_POS 1.0e15f  // zero-mass padding bodies sit far away: they exert and (after masking) receive nothing

// One rotation step: IPT unordered pairs per lane, written stage by stage over all IPT pairs so
// that every instruction's inputs were produced at least IPT instructions earlier (the compiler's
// own schedule chains dependent fma -> rsq -> mul -> fma back to back and issues at ~4 cycles per
// instruction instead of ~2.3; sched_barrier keeps the stages in this order).
template <int IPT, int DBG = 0, bool SYM = true>
__device__ __forceinline__ void pair_evals(const float (&xi)[IPT], const float (&yi)[IPT], const float (&zi)[IPT],
                                           const float (&mi)[IPT], float (&axi)[IPT], float (&ayi)[IPT],
                                           float (&azi)[IPT], float xj, float yj, float zj, float mj, float& axj,
                                           float& ayj, float& azj, float eps2) {
    float dx[IPT], dy[IPT], dz[IPT], r[IPT], sj[IPT];
#pragma unroll
    for (int q = 0; q < IPT; ++q) dx[q] = xj - xi[q];
#pragma unroll
    for (int q = 0; q < IPT; ++q) dy[q] = yj - yi[q];
#pragma unroll
    for (int q = 0; q < IPT; ++q) dz[q] = zj - zi[q];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < IPT; ++q) r[q] = __builtin_fmaf(dx[q], dx[q], eps2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < IPT; ++q) r[q] = __builtin_fmaf(dy[q], dy[q], r[q]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < IPT; ++q) r[q] = __builtin_fmaf(dz[q], dz[q], r[q]);
    __builtin_amdgcn_sched_barrier(0);
    // v_rsq_f32 costs ~8 cycles back to back but ~13-15 in this mixed stream (measured in place by
    // swapping it for a multiply); raising the wave's priority around the burst does not help
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
        if (DBG & 2) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(r[q]));  // timing experiment: no transcendental
        else r[q] = __builtin_amdgcn_rsqf(r[q]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < IPT; ++q) sj[q] = r[q] * r[q];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < IPT; ++q) r[q] = sj[q] * r[q];   // rinv^3
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < IPT; ++q) sj[q] = mj * r[q];     // what body j does to body i
    __builtin_amdgcn_sched_barrier(0);
    if (SYM) {
#pragma unroll
        for (int q = 0; q < IPT; ++q) r[q] = mi[q] * r[q];   // what body i does to body j
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < IPT; ++q) {
            axi[q] = __builtin_fmaf(dx[q], sj[q], axi[q]);
            axj = __builtin_fmaf(-dx[q], r[q], axj);
            ayi[q] = __builtin_fmaf(dy[q], sj[q], ayi[q]);
            ayj = __builtin_fmaf(-dy[q], r[q], ayj);
            azi[q] = __builtin_fmaf(dz[q], sj[q], azi[q]);
            azj = __builtin_fmaf(-dz[q], r[q], azj);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {  // one-sided: only the resident bodies are updated
#pragma unroll
        for (int q = 0; q < IPT; ++q) axi[q] = __builtin_fmaf(dx[q], sj[q], axi[q]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < IPT; ++q) ayi[q] = __builtin_fmaf(dy[q], sj[q], ayi[q]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < IPT; ++q) azi[q] = __builtin_fmaf(dz[q], sj[q], azi[q]);
        __builtin_amdgcn_sched_barrier(0);
    }
}


// ---- packed-fp32 form --------------------------------------------------------------------------
// Two resident bodies per v_pk_* instruction.  Per wave-instruction v_pk_fma_f32 costs ~4.2-4.4 SIMD
// cycles at any occupancy >= 2 waves/SIMD, two v_fma_f32 cost 2 x 2.7 at the 3 waves/SIMD these
// kernels run at (tools/microbench_valu.hip): the same work in ~20 % fewer issue cycles.  The
// travelling body's accumulators become two-wide partial sums (one per half); both halves rotate
// with the body and are added when the chunk is written out.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f splat2(float v) { v2f r = {v, v}; return r; }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

template <int IPT2, int DBG = 0, bool SYM = true>
__device__ __forceinline__ void pair_evals_pk(const v2f (&xi)[IPT2], const v2f (&yi)[IPT2], const v2f (&zi)[IPT2],
                                              const v2f (&mi)[IPT2], v2f (&axi)[IPT2], v2f (&ayi)[IPT2],
                                              v2f (&azi)[IPT2], float xj, float yj, float zj, float mj, v2f& axj,
                                              v2f& ayj, v2f& azj, float eps2) {
    v2f dx[IPT2], dy[IPT2], dz[IPT2], r[IPT2], sj[IPT2];
    const v2f xj2 = splat2(xj), yj2 = splat2(yj), zj2 = splat2(zj), mj2 = splat2(mj), e2 = splat2(eps2);
#pragma unroll
    for (int q = 0; q < IPT2; ++q) dx[q] = xj2 - xi[q];
#pragma unroll
    for (int q = 0; q < IPT2; ++q) dy[q] = yj2 - yi[q];
#pragma unroll
    for (int q = 0; q < IPT2; ++q) dz[q] = zj2 - zi[q];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < IPT2; ++q) r[q] = pk_fma(dx[q], dx[q], e2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < IPT2; ++q) r[q] = pk_fma(dy[q], dy[q], r[q]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < IPT2; ++q) r[q] = pk_fma(dz[q], dz[q], r[q]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < IPT2; ++q) {
        if (DBG & 2) { asm volatile("v_mul_f32 %0, %0, %0" : "+v"(r[q].x)); asm volatile("v_mul_f32 %0, %0, %0" : "+v"(r[q].y)); }
        else { r[q].x = __builtin_amdgcn_rsqf(r[q].x); r[q].y = __builtin_amdgcn_rsqf(r[q].y); }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < IPT2; ++q) sj[q] = r[q] * r[q];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < IPT2; ++q) r[q] = sj[q] * r[q];   // rinv^3
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < IPT2; ++q) sj[q] = mj2 * r[q];    // what body j does to bodies i
    __builtin_amdgcn_sched_barrier(0);
    if (SYM) {
#pragma unroll
        for (int q = 0; q < IPT2; ++q) r[q] = mi[q] * r[q];   // what bodies i do to body j
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < IPT2; ++q) {
            axi[q] = pk_fma(dx[q], sj[q], axi[q]);
            axj = pk_fma(-dx[q], r[q], axj);
            ayi[q] = pk_fma(dy[q], sj[q], ayi[q]);
            ayj = pk_fma(-dy[q], r[q], ayj);
            azi[q] = pk_fma(dz[q], sj[q], azi[q]);
            azj = pk_fma(-dz[q], r[q], azj);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {  // one-sided: only the resident bodies are updated
#pragma unroll
        for (int q = 0; q < IPT2; ++q) axi[q] = pk_fma(dx[q], sj[q], axi[q]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < IPT2; ++q) ayi[q] = pk_fma(dy[q], sj[q], ayi[q]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < IPT2; ++q) azi[q] = pk_fma(dz[q], sj[q], azi[q]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// scalar register arrays <-> packed pairs (bodies 2q and 2q+1 share a register pair)
template <int IPT>
__device__ __forceinline__ void pack_pairs(const float (&a)[IPT], v2f (&b)[IPT / 2]) {
#pragma unroll
    for (int q = 0; q < IPT / 2; ++q) b[q] = v2f{a[2 * q], a[2 * q + 1]};
}
template <int IPT>
__device__ __forceinline__ void unpack_pairs(const v2f (&b)[IPT / 2], float (&a)[IPT]) {
#pragma unroll
    for (int q = 0; q < IPT / 2; ++q) { a[2 * q] = b[q].x; a[2 * q + 1] = b[q].y; }
}

}  // namespace
}  // namespace nbody
