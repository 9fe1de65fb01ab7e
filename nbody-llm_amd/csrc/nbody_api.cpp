// nbody_api.cpp -- the C ABI of include/nbody_hip.h: host orchestration of one shard.
//
// One handle = one reference `Simulation` object (src/shared.rs:80-97) living on one GPU.
// step_by follows brute_force.rs:84-90 / barnes_hut.rs:265-271 kernel by kernel:
//   K1 drift_half -> K4 compact (retain) -> [exchange] -> K2 | (host octree + K5) -> K3 kick_drift
// Everything is enqueued on the handle's own stream; the brute-force path never synchronises
// with the host inside nbody_steps, the Barnes-Hut path must (the octree is built on the host).
#include "nbody_handle.h"
#include "nbody_f64.h"
#include "nbody_let.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>


static thread_local std::string g_create_err;

namespace {

using clk = std::chrono::steady_clock;
inline double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

int fail(NbodyHandle* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_err = msg;
    return code;
}

#define HIP_TRY(h, expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(h, NBODY_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

#define TP_TRY(h, expr)                                                                               \
    do {                                                                                              \
        int r_ = (expr);                                                                              \
        if (r_ != NBODY_OK) return fail(h, r_, std::string(#expr) + ": " + (h)->tp->error());         \
    } while (0)

void* pinned_alloc(size_t n) {
    void* p = nullptr;
    if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void pinned_free(void* p) { (void)hipHostFree(p); }

int use_device(NbodyHandle* h) {
    nbody::bind_tuning(&h->tune);   // the launchers below read this handle's knobs (kernels.h)
    HIP_TRY(h, hipSetDevice(h->device));
    return NBODY_OK;
}

void compute_bounds(NbodyHandle* h) {
    float hw = h->width * 0.5f;  // Bounds::new
    for (int i = 0; i < 3; ++i) {
        h->bnd.lo[i] = h->center[i] + (-hw);  // add_scalar(-half_width), shared.rs:224
        h->bnd.hi[i] = h->center[i] + hw;     // shared.rs:228
    }
}

int ensure_aos(NbodyHandle* h, size_t records) {
    if (records <= h->aos_cap) return NBODY_OK;
    if (h->d_aos) (void)hipFree(h->d_aos);
    if (h->h_aos) (void)hipHostFree(h->h_aos);
    h->d_aos = nullptr; h->h_aos = nullptr; h->aos_cap = 0;
    HIP_TRY(h, hipMalloc(&h->d_aos, records * 10 * sizeof(float)));
    HIP_TRY(h, hipHostMalloc(&h->h_aos, records * 10 * sizeof(float), hipHostMallocDefault));
    h->aos_cap = records;
    return NBODY_OK;
}

// refresh the host view of the own count (one 4-byte D2H + sync), only when it may be stale
int sync_count(NbodyHandle* h) {
    if (!h->count_dirty) return NBODY_OK;
    HIP_TRY(h, hipMemcpyAsync(h->h_counts, h->sh.seg_count, sizeof(int) * h->sh.n_seg, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int s = 0; s < h->sh.n_seg; ++s) h->seg_count_host[s] = h->h_counts[s];
    h->n_local = size_t(h->h_counts[h->sh.my_seg]);
    h->count_dirty = false;
    return NBODY_OK;
}

int push_counts(NbodyHandle* h) {
    for (int s = 0; s < h->sh.n_seg; ++s) h->h_counts[s] = h->seg_count_host[s];
    HIP_TRY(h, hipMemcpyAsync(h->sh.seg_count, h->h_counts, sizeof(int) * h->sh.n_seg, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // h_counts is reused
    return NBODY_OK;
}

size_t total_upper(const NbodyHandle* h) {
    size_t t = 0;
    for (int c : h->seg_count_host) t += size_t(c);
    return t;
}

// the once-per-step exchange of half-drifted positions: an in-place
// all-gather of the own segment into every rank's pos_all, plus the live counts
int exchange_begin(NbodyHandle* h) {
    if (h->sh.n_seg == 1 && !h->comm_ready) return NBODY_OK;  // (a 1-rank communicator still runs the collective)
    if (!h->comm_ready) return fail(h, NBODY_ERR_COMM, "world_size > 1 but nbody_comm_init has not been called");
    // comm stream: after the drift/compaction of this step, beside whatever the compute stream does next
    HIP_TRY(h, hipEventRecord(h->ev_drifted, h->stream));
    HIP_TRY(h, hipStreamWaitEvent(h->comm_stream, h->ev_drifted, 0));
    TP_TRY(h, h->tp->group_begin());
    TP_TRY(h, h->tp->all_gather(h->sh.pos_all, size_t(h->sh.seg_cap) * sizeof(float4), h->comm_stream));
    TP_TRY(h, h->tp->all_gather(h->sh.seg_count, sizeof(int), h->comm_stream));
    TP_TRY(h, h->tp->group_end());
    HIP_TRY(h, hipEventRecord(h->ev_gathered, h->comm_stream));
    h->exchange_in_flight = true;
    return NBODY_OK;
}

// everything enqueued on the compute stream after this call sees the gathered positions
int exchange_wait(NbodyHandle* h) {
    if (!h->exchange_in_flight) return NBODY_OK;
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_gathered, 0));
    h->exchange_in_flight = false;
    return NBODY_OK;
}

// anything the transport noticed behind the host's back (a device-side wait that ran out of time, a peer that gave up)
int comm_check(NbodyHandle* h) {
    if (!h->tp) return NBODY_OK;
    int rc = h->tp->check();
    return rc ? fail(h, rc, h->tp->error()) : NBODY_OK;
}

// ---- Vec::push / Vec::swap_remove on a world of index-block shards (collective: every rank makes the same call).
// The global vector is the concatenation of the ranks' blocks, so push appends to the LAST rank's block and
// swap_remove(i) moves the world's last body into slot i -- across ranks if they differ (one 40-byte message).
int sharded_counts_exact(NbodyHandle* h) {   // every rank learns every block's live count
    if (!h->comm_ready) return fail(h, NBODY_ERR_COMM, "world_size > 1 but nbody_comm_init has not been called");
    int rc = exchange_wait(h);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    TP_TRY(h, h->tp->all_gather(h->sh.seg_count, sizeof(int), h->stream));
    h->count_dirty = true;
    return sync_count(h);
}

int push_own_count(NbodyHandle* h) {
    h->h_counts[h->sh.my_seg] = int(h->n_local);
    HIP_TRY(h, hipMemcpyAsync(h->sh.own_count(), h->h_counts + h->sh.my_seg, sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return NBODY_OK;
}

int sharded_add_point(NbodyHandle* h, const void* particle) {
    int rc = sharded_counts_exact(h);
    if (rc) return rc;
    const int G = h->sh.n_seg, last = G - 1;
    if (total_upper(h) >= h->cfg.capacity || h->seg_count_host[last] >= h->sh.seg_cap)
        return fail(h, NBODY_ERR_CAPACITY, "capacity exhausted (a push goes to the end of the vector: the last rank's block is full)");
    if (h->sh.my_seg == last) {
        const float* p = static_cast<const float*>(particle);
        float4 rec[3] = {make_float4(p[0], p[1], p[2], p[9]), make_float4(p[3], p[4], p[5], 0.f), make_float4(p[6], p[7], p[8], 0.f)};
        const size_t k = h->n_local;
        HIP_TRY(h, hipMemcpyAsync(h->sh.own_pos() + k, &rec[0], sizeof(float4), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->sh.vel + k, &rec[1], sizeof(float4), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->sh.acc + k, &rec[2], sizeof(float4), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        h->n_local = k + 1;
        rc = push_own_count(h);
        if (rc) return rc;
    }
    h->seg_count_host[last] += 1;   // (every rank: the bound its grids are sized from; the device copy arrives with the next exchange)
    return NBODY_OK;
}

int sharded_remove_point(NbodyHandle* h, size_t index) {
    int rc = sharded_counts_exact(h);
    if (rc) return rc;
    const int G = h->sh.n_seg, me = h->sh.my_seg;
    if (index >= total_upper(h)) return fail(h, NBODY_ERR_INVALID, "swap_remove index out of range");   // Vec::swap_remove panics
    int r = 0, last = G - 1;
    size_t j = index;
    while (j >= size_t(h->seg_count_host[r])) { j -= size_t(h->seg_count_host[r]); ++r; }
    while (h->seg_count_host[last] == 0) --last;
    const size_t tail = size_t(h->seg_count_host[last]) - 1;   // the world's last body: (last, tail)
    if (r == last) {
        if (me == r && j != tail) {
            HIP_TRY(h, hipMemcpyAsync(h->sh.own_pos() + j, h->sh.own_pos() + tail, sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(h->sh.vel + j, h->sh.vel + tail, sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(h->sh.acc + j, h->sh.acc + tail, sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
        }
    } else if (me == last || me == r) {
        rc = ensure_aos(h, 2);
        if (rc) return rc;
        if (me == last) {   // the world's last body as one PointParticle record, to the rank that holds slot `index`
            nbody::launch_soa_to_aos(h->stream, h->d_aos, 10, 1, h->sh.own_pos() + tail, h->sh.vel + tail, h->sh.acc + tail);
            TP_TRY(h, h->tp->send(h->d_aos, 40, r, h->stream));
        } else {
            TP_TRY(h, h->tp->recv(h->d_aos, 40, last, h->stream));
            nbody::launch_aos_to_soa(h->stream, h->d_aos, 10, 1, h->sh.own_pos() + j, h->sh.vel + j, h->sh.acc + j);
        }
        HIP_TRY(h, hipGetLastError());
    }
    if (me == last) {
        h->n_local = tail;
        rc = push_own_count(h);
        if (rc) return rc;
    } else {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    h->seg_count_host[last] -= 1;
    return comm_check(h);
}

struct ForceTimer {  // HIP events around a force-kernel launch, on the launch stream
    NbodyHandle* h;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    explicit ForceTimer(NbodyHandle* hh) : h(hh) {
        h->timed_this = false;
        if (!h->profiling) return;
        // (an event pair costs the stream ~11 us: with nbody_set_profiling(h, k > 1) only every k-th launch is bracketed)
        if (h->profile_every > 1 && (h->profile_tick++ % unsigned(h->profile_every)) != 0) return;
        h->timed_this = true;
        if (!h->ev_free.empty()) { ev = h->ev_free.back(); h->ev_free.pop_back(); }
        else {
            if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) { ev = {nullptr, nullptr}; return; }
        }
        (void)hipEventRecord(ev.first, h->stream);
    }
    ~ForceTimer() {
        if (!ev.first) return;
        (void)hipEventRecord(ev.second, h->stream);
        h->ev_pending.push_back(ev);
    }
};

int drain_events(NbodyHandle* h) {
    if (h->ev_pending.empty()) return NBODY_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (auto& ev : h->ev_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
            h->stats.force_kernel_ms += ms;
            h->stats.force_launches += 1;
        }
        h->ev_free.push_back(ev);
    }
    h->ev_pending.clear();
    return NBODY_OK;
}

constexpr size_t kShardedSymMinBodies = 2048; // sharded: own-own symmetric + remote one-sided from this size up

// (re)build the symmetric kernel's plan when the number of resident sets changes
int ensure_sym_plan(NbodyHandle* h) {
    const size_t set = size_t(64) * size_t(nbody::sym_bodies_per_lane(std::max<size_t>(1, h->n_local)));   // bodies of a resident set (make_sym_plan)
    const int A = int((std::max<size_t>(1, h->n_local) + set - 1) / set);  // (an empty shard still plans one set)
    const int knobs = nbody::tuning().sym_wpb * 100 + nbody::tuning().sym_rounds + nbody::tuning().sym_k * 10000 + nbody::tuning().sym_ipt * 1000000;
    if (h->sym_plan.A == A && h->sym_plan.ipt * 64 == int(set) && h->d_sym_bounds && h->sym_waves == knobs) return NBODY_OK;
    h->sym_waves = knobs;
    h->sym_plan = nbody::make_sym_plan(int(std::max<size_t>(1, h->n_local)));
    nbody::SymPlan& p = h->sym_plan;
    h->cross_on = false;
    if (h->sh.n_seg > 1) {
        const size_t cap_pad = (size_t(h->sh.seg_cap) + 63) / 64 * 64;
        p.plane_stride = std::max(p.n_pad, cap_pad);
        if (nbody::tuning().cross_sym && h->sh.n_seg <= 2 * (nbody::CrossPartners::kMax - 1)) {
            // every unordered pair between shards once: this GPU is resident for some partners and
            // receives the partial sums the others accumulated for its bodies
            h->cross = nbody::make_cross_plan(h->sh.my_seg, h->sh.n_seg, h->sh.seg_cap, int(std::max<size_t>(1, h->n_local)));
            h->cross_on = true;
            h->recv_plane0 = p.n_planes + h->cross.k_res;
            p.n_planes += h->cross.k_res + h->cross.n_recv;
        } else {  // one-sided planes for the other shards' bodies: CU-sized 12-wave workgroups
            const int a_os = int((p.n_pad + 511) / 512);   // (k_bf_os keeps resident sets of 512 bodies whatever the symmetric kernel's are)
            p.k_os = std::max(1, std::min(256, 3072 / std::max(1, a_os)));
            p.plane_stride = std::max(p.plane_stride, size_t(a_os) * 512);   // (its padded lanes write their rows too)
            p.n_planes += p.k_os;
        }
    }
    if (!h->d_sym_bounds) HIP_TRY(h, hipMalloc(&h->d_sym_bounds, 128 * sizeof(int)));
    HIP_TRY(h, hipMemcpyAsync(h->d_sym_bounds, p.bounds.data(), p.bounds.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // p.bounds is pageable
    const size_t need = size_t(p.n_planes) * p.plane_stride;
    if (need > h->planes_cap) {
        if (h->d_planes) (void)hipFree(h->d_planes);
        h->d_planes = nullptr; h->planes_cap = 0;
        HIP_TRY(h, hipMalloc(&h->d_planes, need * sizeof(float4)));
        h->planes_cap = need;
    }
    if (h->cross_on) {
        const nbody::CrossPlan& c = h->cross;
        if (c.slices.size() > h->cross_slices_cap) {
            if (h->d_cross_slices) (void)hipFree(h->d_cross_slices);
            h->d_cross_slices = nullptr; h->cross_slices_cap = 0;
            HIP_TRY(h, hipMalloc(&h->d_cross_slices, (c.slices.size() + 64) * sizeof(int4)));
            h->cross_slices_cap = c.slices.size() + 64;
        }
        if (!c.slices.empty()) {
            HIP_TRY(h, hipMemcpyAsync(h->d_cross_slices, c.slices.data(), c.slices.size() * sizeof(int4), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
        }
        const size_t xneed = size_t(std::max(1, c.parts.n)) * size_t(c.A) * p.plane_stride;
        if (xneed > h->xplanes_cap) {
            if (h->d_xplanes) (void)hipFree(h->d_xplanes);
            h->d_xplanes = nullptr; h->xplanes_cap = 0;
            HIP_TRY(h, hipMalloc(&h->d_xplanes, xneed * sizeof(float4)));
            h->xplanes_cap = xneed;
        }
        const size_t sneed = size_t(std::max(1, c.parts.n)) * p.plane_stride;
        if (sneed > h->send_cap) {
            if (h->d_send) (void)hipFree(h->d_send);
            h->d_send = nullptr; h->send_cap = 0;
            HIP_TRY(h, hipMalloc(&h->d_send, sneed * sizeof(float4)));
            h->send_cap = sneed;
        }
    }
    return NBODY_OK;
}

int bf_forces(NbodyHandle* h) {
    const float eps2 = h->g_soft * h->g_soft;  // brute_force.rs:69
    const bool fast = h->cfg.math_mode == NBODY_MATH_FAST && nbody::tuning().bf_fast_variant == 0;
    const bool sharded = h->sh.n_seg > 1;
    // Sharded: decided from the shard CAPACITY, which every rank shares.  The live counts differ from rank
    // to rank (ragged last block, bodies leaving the box), and a rank that chose another scheme than its
    // peers would neither send nor expect the partial sums the others exchange with it (a hang in RCCL).
    const bool sym = fast && (sharded ? size_t(h->sh.seg_cap) >= kShardedSymMinBodies : h->n_local >= size_t(std::max(1024, nbody::tuning().sym_min_bodies)));
    const size_t tot = total_upper(h);
    if (sym) {
        int rc = ensure_sym_plan(h);
        if (rc) return rc;
        if (h->sym_pairs_n != h->n_local) {
            h->sym_pairs = nbody::sym_main_pairs(h->sym_plan, h->n_local);
            h->sym_pairs_n = h->n_local;
        }
    }
    const nbody::SymPlan& p = h->sym_plan;
    uint64_t timed = 0;  // interactions of the launch the HIP events bracket (the dominant one)
    if (!(sym && sharded)) {
        int rc = exchange_wait(h);
        if (rc) return rc;
    }
    if (!sym) {
        ForceTimer t(h);
        if (h->cfg.math_mode == NBODY_MATH_STRICT) nbody::launch_bf_forces_strict(h->stream, h->sh, int(h->n_local), h->g, eps2);
        else nbody::launch_bf_forces_fast(h->stream, h->sh, int(h->n_local), h->g, eps2);
        timed = tot > 0 ? uint64_t(h->n_local) * uint64_t(tot - 1) : 0;
    } else if (!sharded) {
        {
            ForceTimer t(h);
            nbody::launch_bf_sym_main(h->stream, h->sh, p, h->d_sym_bounds, h->d_planes, int(h->n_local), eps2);
        }
        timed = 2 * h->sym_pairs;
        h->tail_pending = true;
    } else {
        // own shard symmetric (needs no remote data: it overlaps the exchange of positions) ...
        nbody::launch_bf_sym_main(h->stream, h->sh, p, h->d_sym_bounds, h->d_planes, int(h->n_local), eps2);
        {
            int rc = exchange_wait(h);
            if (rc) return rc;
        }
        if (h->cross_on) {
            // ... then the pairs with the partner shards, both sides; the partial sums for their bodies
            // go back to their owners before the planes are added up (forces_finish)
            const nbody::CrossPlan& c = h->cross;
            {
                ForceTimer t(h);
                nbody::launch_bf_cross(h->stream, h->sh, c, h->d_cross_slices,
                                       h->d_planes + size_t(h->recv_plane0 - c.k_res) * p.plane_stride, h->d_xplanes,
                                       h->d_send, p.plane_stride, eps2);
            }
            for (int i = 0; i < c.parts.n; ++i) {  // unordered pairs x 2, from the host's (upper-bound) counts
                const long long set = 64LL * c.ipt;
                const long long own = std::max(0LL, std::min<long long>(h->n_local, set * c.parts.a1[i]) - set * c.parts.a0[i]);
                const long long theirs = std::max(0LL, std::min<long long>(h->seg_count_host[c.parts.seg[i]], 64LL * c.parts.c1[i]) - 64LL * c.parts.c0[i]);
                timed += 2ull * uint64_t(own) * uint64_t(theirs);
            }
        } else {
            // ... then the other shards' bodies one-sided
            ForceTimer t(h);
            nbody::launch_bf_os(h->stream, h->sh, int((p.n_pad + 511) / 512), p.k_os, h->d_planes + size_t(p.n_planes - p.k_os) * p.plane_stride, p.plane_stride, eps2);
            timed = uint64_t(h->n_local) * uint64_t(tot - h->n_local);
        }
        h->tail_pending = true;
    }
    HIP_TRY(h, hipGetLastError());
    // (NbodyStats::interactions is counted on the device from the live counts: Shard::inter)
    if (tot > 0 && h->timed_this) h->stats.force_kernel_interactions += timed;
    return NBODY_OK;
}

int ensure_tree_dev(NbodyHandle* h, size_t nodes, size_t order) {
    if (nodes > h->d_node_cap) {
        if (h->d_nodes) (void)hipFree(h->d_nodes);
        h->d_nodes = nullptr; h->d_node_cap = 0;
        size_t cap = nodes + nodes / 4 + 1024;
        HIP_TRY(h, hipMalloc(&h->d_nodes, cap * 2 * sizeof(float4)));
        h->d_node_cap = cap;
    }
    if (order > h->d_order_cap) {
        if (h->d_order) (void)hipFree(h->d_order);
        h->d_order = nullptr; h->d_order_cap = 0;
        size_t cap = order + order / 4 + 1024;
        HIP_TRY(h, hipMalloc(&h->d_order, cap * sizeof(int)));
        h->d_order_cap = cap;
    }
    return NBODY_OK;
}

// BarnesHutSimulation::update_forces (barnes_hut.rs:250-263): rebuild the tree from the current
// positions, then one walk per body.
int bh_walk_device_tree(NbodyHandle* h, bool* fell_back);
int bh_walk_device_tree_async(NbodyHandle* h);
int resolve_async(NbodyHandle* h);
int step_end(NbodyHandle* h, float dt);
int step_impl(NbodyHandle* h, float dt);

// strict math with the reference leaf rule walks with the reference's nested sums (bit-exact): per own body a stack
// of the open cells on its path, 16 bytes each -- as many levels as the tree is deep (the device build stops at 42;
// the host build reports its depth), not NBODY_MAX_TREE_DEPTH of them (13 GB at N = 2^22)
int ensure_nested_stack(NbodyHandle* h, nbody::TreeDev* td) {
    if (h->cfg.math_mode != NBODY_MATH_STRICT || h->cfg.leaf_mode != NBODY_LEAF_REFERENCE) return NBODY_OK;
    const size_t lanes = (size_t(h->sh.seg_cap) + 255) / 256 * 256;
    const int levels = (h->tree_on_device ? 43 : h->tree.max_depth) + 2;   // (the device build goes to 42 levels)
    if (lanes > h->nested_cap || levels > h->nested_levels) {
        if (h->d_nested_stack) (void)hipFree(h->d_nested_stack);
        h->d_nested_stack = nullptr; h->nested_cap = 0; h->nested_levels = 0;
        const int lv = std::max(levels + 8, 32);
        HIP_TRY(h, hipMalloc(&h->d_nested_stack, lanes * size_t(lv) * sizeof(float4)));
        h->nested_cap = lanes;
        h->nested_levels = lv;
    }
    td->nested_stack = h->d_nested_stack;
    td->nested_stride = h->nested_cap;
    return NBODY_OK;
}

// fast math, variant 3: buffers of the LDS-staged walk and the threshold that picks the staged nodes.
// Called after this step's first host synchronisation, so h_hot_info holds the previous pass's flagged count.
int setup_lds_walk(NbodyHandle* h, nbody::TreeDev* td, size_t n_tree) {
#ifndef NBODY_TUNING
    (void)h; (void)td; (void)n_tree;   // (the experimental walks live in the tuning build)
    return NBODY_OK;
#else
    // (its stack holds the 42 levels of the device build; a deeper host-built tree is walked by k_bh_walk)
    if (h->cfg.math_mode == NBODY_MATH_FAST && nbody::tuning().bh_walk_variant == 5 && (h->tree_on_device || h->tree.max_depth <= 42)) {
        if (h->bfs_cap < h->d_node_cap) {
            if (h->d_bfs) (void)hipFree(h->d_bfs);
            if (h->d_bfs_ws) (void)hipFree(h->d_bfs_ws);
            h->d_bfs = nullptr; h->d_bfs_ws = nullptr; h->bfs_cap = 0;
            HIP_TRY(h, hipMalloc(&h->d_bfs, h->d_node_cap * 2 * sizeof(float4)));
            HIP_TRY(h, hipMalloc(&h->d_bfs_ws, nbody::bfs_workspace_bytes(h->d_node_cap)));
            h->bfs_cap = h->d_node_cap;
        }
        td->bfs = h->d_bfs; td->bfs_ws = h->d_bfs_ws; td->bfs_cap = h->bfs_cap;
        return NBODY_OK;
    }
    if (h->cfg.math_mode != NBODY_MATH_FAST || nbody::tuning().bh_walk_variant != 3 || nbody::tuning().bh_hot_cap <= 0) return NBODY_OK;
    const int M = std::min(nbody::tuning().bh_hot_cap, 5000);  // 160 KB of LDS per CU, 32 B per record
    if (h->walk_cap < h->d_node_cap) {
        if (h->d_walk) (void)hipFree(h->d_walk);
        if (h->d_unified) (void)hipFree(h->d_unified);
        h->d_walk = nullptr; h->d_unified = nullptr; h->walk_cap = 0;
        HIP_TRY(h, hipMalloc(&h->d_walk, (h->d_node_cap + 1) * 2 * sizeof(float4)));
        HIP_TRY(h, hipMalloc(&h->d_unified, (h->d_node_cap + 1) * sizeof(int)));
        h->walk_cap = h->d_node_cap;
    }
    if (h->hot_cap != M) {
        if (h->d_hot) (void)hipFree(h->d_hot);
        h->d_hot = nullptr; h->hot_cap = 0;
        HIP_TRY(h, hipMalloc(&h->d_hot, size_t(M) * 2 * sizeof(float4)));
        HIP_TRY(h, hipMemsetAsync(h->d_hot, 0, size_t(M) * 2 * sizeof(float4), h->stream));
        h->hot_cap = M;
        h->hot_threshold_n = 0;
    }
    if (!h->d_hot_info) {
        HIP_TRY(h, hipMalloc(&h->d_hot_info, 2 * sizeof(int)));
        HIP_TRY(h, hipMemsetAsync(h->d_hot_info, 0, 2 * sizeof(int), h->stream));
        HIP_TRY(h, hipHostMalloc(&h->h_hot_info, 2 * sizeof(int), hipHostMallocDefault));
        h->h_hot_info[0] = h->h_hot_info[1] = 0;
    }
    if (h->hot_threshold_n == 0 || n_tree > 2 * h->hot_threshold_n || 2 * n_tree < h->hot_threshold_n) {
        // first guess: the grandparent holds 1/64 of the bodies (2 500 nodes at N = 65 536 Plummer)
        h->hot_threshold = int(std::max<size_t>(8, n_tree / 64));
        h->hot_threshold_n = std::max<size_t>(1, n_tree);
    } else {
        const int flagged = h->h_hot_info[1];
        if (flagged > M) h->hot_threshold = h->hot_threshold + h->hot_threshold / 8 + 1;
        else if (flagged < M - M / 3 && h->hot_threshold > 2) h->hot_threshold = h->hot_threshold - h->hot_threshold / 8 - 1;
    }
    td->walk = h->d_walk; td->unified = h->d_unified; td->hot = h->d_hot; td->hot_info = h->d_hot_info;
    td->hot_cap = M; td->hot_threshold = h->hot_threshold;
    return NBODY_OK;
#endif
}

int bh_forces(NbodyHandle* h) {
    Shard& sh = h->sh;
    {
        int rc = exchange_wait(h);
        if (rc) return rc;
    }
    h->last_step_async = false;
    if (h->cfg.tree_build == NBODY_TREE_DEVICE && !h->host_tree_once) {
        // no read-back at all: single shard, the plain or the strict walk (the experimental walks want the node count)
        const bool plain_walk = h->cfg.math_mode == NBODY_MATH_STRICT || nbody::tuning().bh_walk_variant == 0;
        if (h->async_bh && plain_walk && !nbody::tuning().bh_walk_debug) return bh_walk_device_tree_async(h);
        int rc = resolve_async(h);
        if (rc) return rc;
        bool fell_back = false;
        rc = bh_walk_device_tree(h, &fell_back);
        if (rc || !fell_back) return rc;
        // deeper than 42 levels somewhere: this step's tree comes from the host build below
    }
    h->host_tree_once = false;
    h->tree_on_device = false;
    auto t0 = clk::now();
    // positions of every segment (upper-bound counts) + the live counts, one sync
    for (int s = 0; s < sh.n_seg; ++s) {
        size_t cnt = size_t(h->seg_count_host[s]);
        if (cnt)
            HIP_TRY(h, hipMemcpyAsync(h->h_pos + 4 * size_t(s) * sh.seg_cap, sh.pos_all + size_t(s) * sh.seg_cap,
                                      cnt * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, hipMemcpyAsync(h->h_counts, sh.seg_count, sizeof(int) * sh.n_seg, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int s = 0; s < sh.n_seg; ++s) h->seg_count_host[s] = h->h_counts[s];
    h->n_local = size_t(h->h_counts[sh.my_seg]);
    h->count_dirty = false;
    double copy_ms = ms_since(t0);

    auto t1 = clk::now();
    nbody::build_octree(h->h_pos, sh.n_seg, sh.seg_cap, h->h_counts, h->center, h->width, *h->pool, h->tree_scratch, h->tree);
    if (h->tree.too_deep) return fail(h, NBODY_ERR_TREE_DEPTH, "octree deeper than NBODY_MAX_TREE_DEPTH (coincident bodies?)");
    // bodies of the own segment in tree order (ids are s*seg_cap + j)
    const int32_t* order = h->tree.order;
    size_t n_order = h->tree.n_order;
    if (sh.n_seg > 1) {
        h->own_order.clear();
        const int lo = sh.my_seg * sh.seg_cap, hi = lo + sh.seg_cap;
        for (size_t k = 0; k < h->tree.n_order; ++k) {
            int id = h->tree.order[k];
            if (id >= lo && id < hi) h->own_order.push_back(id - lo);
        }
        order = h->own_order.data();
        n_order = h->own_order.size();
    }
    h->stats.tree_build_ms += ms_since(t1);
    h->stats.tree_nodes = h->tree.n_nodes;

    auto t2 = clk::now();
    int rc = ensure_tree_dev(h, h->tree.n_nodes, n_order);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->d_nodes, h->tree.nodes, h->tree.n_nodes * sizeof(nbody::NodeRec), hipMemcpyHostToDevice, h->stream));
    if (n_order) HIP_TRY(h, hipMemcpyAsync(h->d_order, order, n_order * sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (sh.n_seg > 1) HIP_TRY(h, hipStreamSynchronize(h->stream));  // own_order is pageable and reused
    h->stats.tree_copy_ms += copy_ms + ms_since(t2);

    nbody::TreeDev td;
    td.nodes = h->d_nodes; td.n_nodes = int(h->tree.n_nodes);
    td.order = h->d_order; td.n_order = int(n_order);
    // split the node range over several waves per body group when there are too few bodies to fill
    // the chip (>= 8 waves per SIMD wanted: the walk is bound by the latency of dependent loads)
    {
        constexpr int kMaxSplit = 64, kMaxAnc = 192;
        // ~3 waves per wave slot of the chip (256 CUs x 32), handed out heaviest first (nbody::tuning().bh_walk_order): the launch
        // lasts as long as its slowest wave, and smaller pieces started in the right order shorten that tail
        // (N = 65 536: 24 segments 0.310 ms, 8 segments 0.336 ms; tools/tune_bh_order.py)
        int K = nbody::walk_plan(n_order, h->cfg.math_mode != NBODY_MATH_STRICT, kMaxSplit, h->theta2).segments;
        // strict math is the parity path: one segment, so every lane adds in the reference's order (bit-exact)
        if (h->cfg.math_mode == NBODY_MATH_STRICT && nbody::tuning().bh_walk_split <= 0) K = 1;
        while (nbody::tuning().bh_walk_split <= 0 && K > 1 && size_t(K) * 16 > h->tree.n_nodes) K /= 2;   // (a pinned split is taken as given)
        if (K > 1) {
            if (!h->d_split) {
                HIP_TRY(h, hipMalloc(&h->d_split, (kMaxSplit + 1 + kMaxSplit + kMaxSplit * kMaxAnc) * sizeof(int)));
                HIP_TRY(h, hipHostMalloc(&h->h_split, (kMaxSplit + 1 + kMaxSplit + kMaxSplit * kMaxAnc) * sizeof(int), hipHostMallocDefault));
            }
            int* first = h->h_split;
            int* n_anc = first + kMaxSplit + 1;
            int* anc = n_anc + kMaxSplit;
            const nbody::NodeRec* nd = h->tree.nodes;
            const int nn = int(h->tree.n_nodes);
            for (int k = 0; k <= K; ++k) first[k] = int((long long)nn * k / K);
            for (int k = 0; k < K; ++k) {  // ancestors of first[k]: walk down from the root along the skip links
                int cnt = 0, j = 0;
                const int target = first[k];
                while (j != target && cnt < kMaxAnc) {
                    anc[k * kMaxAnc + cnt++] = j;          // j < target < skip(j): an ancestor
                    int c = j + 1;                         // its first child
                    while (nd[c].b.skip <= target) c = nd[c].b.skip;  // siblings in orthant order
                    j = c;
                }
                n_anc[k] = cnt;
            }
            const size_t ints = size_t(kMaxSplit + 1 + kMaxSplit + K * kMaxAnc);
            HIP_TRY(h, hipMemcpyAsync(h->d_split, h->h_split, ints * sizeof(int), hipMemcpyHostToDevice, h->stream));
            const size_t need = size_t(K) * sh.seg_cap;
            if (need > h->walk_planes_cap) {
                if (h->d_walk_planes) (void)hipFree(h->d_walk_planes);
                h->d_walk_planes = nullptr; h->walk_planes_cap = 0;
                HIP_TRY(h, hipMalloc(&h->d_walk_planes, need * sizeof(float4)));
                h->walk_planes_cap = need;
            }
            td.n_split = K;
            td.split_first = h->d_split;
            td.split_n_anc = h->d_split + kMaxSplit + 1;
            td.split_anc = h->d_split + kMaxSplit + 1 + kMaxSplit;
            td.split_planes = h->d_walk_planes;
            td.split_stride = size_t(sh.seg_cap);
        } else {
            if (!h->d_split) {
                HIP_TRY(h, hipMalloc(&h->d_split, (kMaxSplit + 1 + kMaxSplit + kMaxSplit * kMaxAnc) * sizeof(int)));
                HIP_TRY(h, hipHostMalloc(&h->h_split, (kMaxSplit + 1 + kMaxSplit + kMaxSplit * kMaxAnc) * sizeof(int), hipHostMallocDefault));
            }
            h->h_split[0] = 0; h->h_split[1] = int(h->tree.n_nodes); h->h_split[kMaxSplit + 1] = 0;
            HIP_TRY(h, hipMemcpyAsync(h->d_split, h->h_split, (kMaxSplit + 2) * sizeof(int), hipMemcpyHostToDevice, h->stream));
            td.n_split = 1;
            td.split_first = h->d_split;
            td.split_n_anc = h->d_split + kMaxSplit + 1;
            td.split_anc = h->d_split + kMaxSplit + 1 + kMaxSplit;
        }
    }
    {
        int rc_ns = ensure_nested_stack(h, &td);
        if (rc_ns) return rc_ns;
        rc_ns = setup_lds_walk(h, &td, h->tree.n_order);
        if (rc_ns) return rc_ns;
    }
    {
        ForceTimer t(h);
        int kicked = 0;
        nbody::launch_bh_walk(h->stream, sh, td, h->g, h->g_soft * h->g_soft, h->theta2,
                              h->cfg.math_mode == NBODY_MATH_FAST, h->d_counters, h->cfg.leaf_mode == NBODY_LEAF_DIRECT,
                              h->kick_pending ? &h->kick_dt : nullptr, &kicked);
        if (kicked) h->kick_pending = false;  // the plane reduction applied the kick + half drift
    }
    if (td.hot_cap > 0) HIP_TRY(h, hipMemcpyAsync(h->h_hot_info, h->d_hot_info, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipGetLastError());
    return NBODY_OK;
}

// Barnes-Hut force pass with the octree built on the device (kernels_tree.hip): no positions go to
// the host, no node array comes back; one 8-byte read-back (node count, flags) per step.
int bh_walk_device_tree(NbodyHandle* h, bool* fell_back) {
    Shard& sh = h->sh;
    auto t1 = clk::now();
    const bool sharded = sh.n_seg > 1;
    const size_t n_cap = size_t(sh.seg_cap) * sh.n_seg;  // the tree holds the bodies of every segment
    if (h->tree_ws_cap < n_cap) {
        if (h->d_tree_ws) (void)hipFree(h->d_tree_ws);
        if (h->d_tree_cat) (void)hipFree(h->d_tree_cat);
        h->d_tree_ws = nullptr; h->d_tree_cat = nullptr; h->tree_ws_cap = 0;
        HIP_TRY(h, hipMalloc(&h->d_tree_ws, nbody::tree_build_workspace_bytes(n_cap)));
        if (sharded) HIP_TRY(h, hipMalloc(&h->d_tree_cat, nbody::tree_cat_bytes(n_cap)));
        h->tree_ws_cap = n_cap;
    }
    if (!h->d_tree_info) {
        HIP_TRY(h, hipMalloc(&h->d_tree_info, 4 * sizeof(int)));
        HIP_TRY(h, hipHostMalloc(&h->h_tree_info, 4 * sizeof(int), hipHostMallocDefault));
    }
    const size_t tot_upper = total_upper(h);
    // a Plummer sphere gives ~1.5 nodes per body; 4 per body + the count read-back below catch the rest
    int rc = ensure_tree_dev(h, std::max<size_t>(h->d_node_cap, 4 * tot_upper + 64), tot_upper);
    if (rc) return rc;
    nbody::TreeCat cat;
    const float4* tree_pos = sh.own_pos();
    const int* tree_count = sh.own_count();
    if (sharded) {  // every GPU builds the same tree over the gathered bodies of all segments
        cat = nbody::tree_cat_layout(h->d_tree_cat, n_cap);
        nbody::launch_tree_cat(h->stream, sh, cat);
        tree_pos = cat.pos;
        tree_count = cat.info;
    }
    nbody::TreeDevWork work;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (nbody::build_octree_device(h->stream, tree_pos, tree_count, int(tot_upper), h->center, h->width,
                                       h->d_tree_ws, n_cap, h->d_nodes, int(h->d_node_cap), h->d_order, h->d_tree_info,
                                       &work, nbody::tuning().bh_walk_variant == 3) != 0)
            return fail(h, NBODY_ERR_HIP, "device octree build: rocPRIM call failed");
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipMemcpyAsync(h->h_tree_info, h->d_tree_info, 3 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        if (sharded)
            HIP_TRY(h, hipMemcpyAsync(h->h_counts, sh.seg_count, sizeof(int) * sh.n_seg, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (!sharded) h->h_counts[0] = h->h_tree_info[2];  // one shard: the tree's body count is the live count
        if (!(h->h_tree_info[1] & 2)) break;
        rc = ensure_tree_dev(h, size_t(h->h_tree_info[0]) + 64, tot_upper);  // more nodes than allowed for: grow, rebuild
        if (rc) return rc;
    }
    for (int s = 0; s < sh.n_seg; ++s) h->seg_count_host[s] = h->h_counts[s];
    h->n_local = size_t(h->h_counts[sh.my_seg]);
    h->count_dirty = false;
    if (h->h_tree_info[1] & 5) { *fell_back = true; return NBODY_OK; }   // deeper than 42 levels / a clump beyond the build's sort: host build
    const int n_nodes = h->h_tree_info[0];
    const size_t n_order = h->n_local;             // bodies this GPU walks
    const size_t n_tree = total_upper(h);          // bodies in the tree (now exact)
    const int* d_walk_order = h->d_order;
    if (sharded) {
        if (nbody::launch_tree_own_order(h->stream, h->d_order, cat, int(n_tree), h->d_tree_ws, nbody::tree_build_tmp_bytes(n_cap)) != 0)
            return fail(h, NBODY_ERR_HIP, "device octree build: rocPRIM call failed");
        d_walk_order = cat.own_order;
    }
    h->stats.tree_build_ms += ms_since(t1);
    h->stats.tree_nodes = uint64_t(n_nodes);
    h->tree_on_device = true;
    h->tree.n_nodes = size_t(n_nodes);  // (the host copy is filled on demand by nbody_tree_export)

    constexpr int kMaxSplit = 64, kMaxAnc = 192;
    if (!h->d_split) {
        HIP_TRY(h, hipMalloc(&h->d_split, (kMaxSplit + 1 + kMaxSplit + kMaxSplit * kMaxAnc) * sizeof(int)));
        HIP_TRY(h, hipHostMalloc(&h->h_split, (kMaxSplit + 1 + kMaxSplit + kMaxSplit * kMaxAnc) * sizeof(int), hipHostMallocDefault));
    }
    int K = nbody::walk_plan(n_order, h->cfg.math_mode != NBODY_MATH_STRICT, kMaxSplit, h->theta2).segments;
    if (h->cfg.math_mode == NBODY_MATH_STRICT && nbody::tuning().bh_walk_split <= 0) K = 1;  // parity path: the reference's sum order
    while (nbody::tuning().bh_walk_split <= 0 && K > 1 && K * 16 > n_nodes) K /= 2;
    if (n_order == 0) K = 1;
    nbody::TreeDev td;
    td.nodes = h->d_nodes; td.n_nodes = n_nodes;
    td.order = d_walk_order; td.n_order = int(n_order);
    td.n_split = K;
    td.split_first = h->d_split;
    td.split_n_anc = h->d_split + kMaxSplit + 1;
    td.split_anc = h->d_split + kMaxSplit + 1 + kMaxSplit;
    if (n_tree > 0)
        nbody::launch_tree_split_anc(h->stream, work, int(n_tree), n_nodes, K, h->d_split, h->d_split + kMaxSplit + 1,
                                     h->d_split + kMaxSplit + 1 + kMaxSplit, kMaxAnc);
    if (K > 1) {
        const size_t need = size_t(K) * sh.seg_cap;
        if (need > h->walk_planes_cap) {
            if (h->d_walk_planes) (void)hipFree(h->d_walk_planes);
            h->d_walk_planes = nullptr; h->walk_planes_cap = 0;
            HIP_TRY(h, hipMalloc(&h->d_walk_planes, need * sizeof(float4)));
            h->walk_planes_cap = need;
        }
        td.split_planes = h->d_walk_planes;
        td.split_stride = size_t(sh.seg_cap);
    }
    {
        int rc_ns = ensure_nested_stack(h, &td);
        if (rc_ns) return rc_ns;
        rc_ns = setup_lds_walk(h, &td, n_tree);
        if (rc_ns) return rc_ns;
    }
    {
        ForceTimer t(h);
        int kicked = 0;
        nbody::launch_bh_walk(h->stream, sh, td, h->g, h->g_soft * h->g_soft, h->theta2,
                              h->cfg.math_mode == NBODY_MATH_FAST, h->d_counters, h->cfg.leaf_mode == NBODY_LEAF_DIRECT,
                              h->kick_pending ? &h->kick_dt : nullptr, &kicked);
        if (kicked) h->kick_pending = false;  // the plane reduction applied the kick + half drift
    }
    if (td.hot_cap > 0) HIP_TRY(h, hipMemcpyAsync(h->h_hot_info, h->d_hot_info, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipGetLastError());
    return NBODY_OK;
}

// The same force pass with nothing read back: the node count, the live body count and the build's flags stay on the
// device (d_tree_info); the split points are placed by k_tree_split_anc from the device's node count, the walk takes
// its body count from the device, and a build that needs the host poisons the run (see NbodyHandle::async_bh).
int bh_walk_device_tree_async(NbodyHandle* h) {
    Shard& sh = h->sh;
    auto t1 = clk::now();
    const size_t n_cap = size_t(sh.seg_cap);
    if (h->tree_ws_cap < n_cap) {
        if (h->d_tree_ws) (void)hipFree(h->d_tree_ws);
        h->d_tree_ws = nullptr; h->tree_ws_cap = 0;
        HIP_TRY(h, hipMalloc(&h->d_tree_ws, nbody::tree_build_workspace_bytes(n_cap)));
        h->tree_ws_cap = n_cap;
    }
    if (!h->d_tree_info) {
        HIP_TRY(h, hipMalloc(&h->d_tree_info, 4 * sizeof(int)));
        HIP_TRY(h, hipHostMalloc(&h->h_tree_info, 4 * sizeof(int), hipHostMallocDefault));
    }
    const size_t n_upper = h->n_local;   // an upper bound of the live count
    int rc = ensure_tree_dev(h, std::max<size_t>(h->d_node_cap, 4 * n_upper + 64), n_upper);
    if (rc) return rc;
    if (h->pending.empty()) HIP_TRY(h, hipMemsetAsync(h->d_poison + 1, 0, sizeof(int), h->stream));   // steps completed: counted from here
    constexpr int kMaxSplit = 64, kMaxAnc = 192;
    if (!h->d_split) {
        HIP_TRY(h, hipMalloc(&h->d_split, (kMaxSplit + 1 + kMaxSplit + kMaxSplit * kMaxAnc) * sizeof(int)));
        HIP_TRY(h, hipHostMalloc(&h->h_split, (kMaxSplit + 1 + kMaxSplit + kMaxSplit * kMaxAnc) * sizeof(int), hipHostMallocDefault));
    }
    int K = nbody::walk_plan(n_upper, h->cfg.math_mode != NBODY_MATH_STRICT, kMaxSplit, h->theta2).segments;
    if (h->cfg.math_mode == NBODY_MATH_STRICT && nbody::tuning().bh_walk_split <= 0) K = 1;  // parity path: the reference's sum order
    while (nbody::tuning().bh_walk_split <= 0 && K > 1 && size_t(K) * 16 > n_upper) K /= 2;   // (a tree has at least as many nodes as bodies)
    if (n_upper == 0) K = 1;
    // the walk's split points ride in the build's last launch (they also make a build that needs the host sticky: Shard::poison)
    nbody::TreeSplitReq req;
    req.n_split = K; req.first = h->d_split; req.n_anc = h->d_split + kMaxSplit + 1; req.anc = h->d_split + kMaxSplit + 1 + kMaxSplit;
    req.max_anc = kMaxAnc; req.info = h->d_tree_info; req.poison = h->d_poison;
    nbody::TreeDevWork work;
    if (nbody::build_octree_device(h->stream, sh.own_pos(), sh.own_count(), int(n_upper), h->center, h->width, h->d_tree_ws, n_cap,
                                   h->d_nodes, int(h->d_node_cap), h->d_order, h->d_tree_info, &work, 0, n_upper > 0 ? &req : nullptr) != 0)
        return fail(h, NBODY_ERR_HIP, "device octree build: rocPRIM call failed");
    HIP_TRY(h, hipGetLastError());
    h->stats.tree_build_ms += ms_since(t1);   // (enqueue time: nothing is waited for)
    h->tree_on_device = true;
    nbody::TreeDev td;
    td.nodes = h->d_nodes; td.n_nodes = int(h->d_node_cap);   // (the plain walks end at the split points, not at n_nodes)
    td.order = h->d_order; td.n_order = int(n_upper);
    td.n_order_dev = h->d_tree_info + 2;
    td.poison = h->d_poison;
    td.n_split = K;
    td.split_first = h->d_split;
    td.split_n_anc = h->d_split + kMaxSplit + 1;
    td.split_anc = h->d_split + kMaxSplit + 1 + kMaxSplit;
    if (n_upper == 0)   // (no build was enqueued: the empty root's one segment, as a launch of its own)
        nbody::launch_tree_split_anc(h->stream, work, int(n_upper), int(h->d_node_cap), K, h->d_split, h->d_split + kMaxSplit + 1,
                                     h->d_split + kMaxSplit + 1 + kMaxSplit, kMaxAnc, h->d_tree_info, h->d_poison);
    if (K > 1) {
        const size_t need = size_t(K) * sh.seg_cap;
        if (need > h->walk_planes_cap) {
            if (h->d_walk_planes) (void)hipFree(h->d_walk_planes);
            h->d_walk_planes = nullptr; h->walk_planes_cap = 0;
            HIP_TRY(h, hipMalloc(&h->d_walk_planes, need * sizeof(float4)));
            h->walk_planes_cap = need;
        }
        td.split_planes = h->d_walk_planes;
        td.split_stride = size_t(sh.seg_cap);
    }
    rc = ensure_nested_stack(h, &td);
    if (rc) return rc;
    {
        ForceTimer t(h);
        int kicked = 0;
        nbody::launch_bh_walk(h->stream, sh, td, h->g, h->g_soft * h->g_soft, h->theta2,
                              h->cfg.math_mode == NBODY_MATH_FAST, h->d_counters, h->cfg.leaf_mode == NBODY_LEAF_DIRECT,
                              h->kick_pending ? &h->kick_dt : nullptr, &kicked);
        if (kicked) h->kick_pending = false;  // the plane reduction applied the kick + half drift
    }
    HIP_TRY(h, hipGetLastError());
    h->last_step_async = true;
    h->count_dirty = true;
    return NBODY_OK;
}

// Where did the device get to?  Confirms the steps enqueued without read-back; if a build poisoned the run, finishes
// the failed step with the host build (which handles any depth) and enqueues the rest again.
int resolve_async(NbodyHandle* h) {
    if (!h->async_bh) return NBODY_OK;
    for (int round = 0; round < 1000000; ++round) {
        HIP_TRY(h, hipMemcpyAsync(h->h_poison, h->d_poison, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        if (h->d_tree_info) HIP_TRY(h, hipMemcpyAsync(h->h_poison + 2, h->d_tree_info, 3 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        const int flags = h->h_poison[0], done = h->h_poison[1];
        if (!flags) {
            if (h->tree_on_device && h->d_tree_info) {   // what the last build produced
                h->tree.n_nodes = size_t(h->h_poison[2]);
                h->stats.tree_nodes = uint64_t(h->h_poison[2]);
            }
            h->pending.clear();
            return NBODY_OK;
        }
        // poisoned: steps [0, done) are complete, step `done` has drifted and compacted, nothing after it has run
        std::vector<NbodyHandle::PendingStep> rest;
        if (size_t(done) < h->pending.size()) rest.assign(h->pending.begin() + done, h->pending.end());
        h->pending.clear();
        HIP_TRY(h, hipMemsetAsync(h->d_poison, 0, 2 * sizeof(int), h->stream));
        if (flags & 2) {   // more nodes than the array holds: double it (the bound 4 n + 64 did not hold for this set)
            int rc = ensure_tree_dev(h, 2 * h->d_node_cap + 64, h->n_local);
            if (rc) return rc;
        }
        if (rest.empty()) {   // it was a force pass outside a step (nbody_update_forces): its caller runs it again
            h->host_tree_once = true;
            return NBODY_OK;
        }
        h->elapsed = rest[0].elapsed_before;
        h->stats.steps -= rest.size();
        h->host_tree_once = true;             // (cleared by the force pass that uses it)
        int rc = sync_count(h);
        if (rc) return rc;
        rc = step_end(h, rest[0].dt);         // forces on the host-built tree, kick + half drift
        if (rc) return rc;
        for (size_t k = 1; k < rest.size(); ++k) {
            rc = step_impl(h, rest[k].dt);    // (enqueued without read-back again: may poison again -> next round)
            if (rc) return rc;
        }
    }
    return fail(h, NBODY_ERR_INVALID, "resolve_async did not converge");
}

// ---- the partial sums other GPUs accumulated for the own bodies (symmetric scheme across shards)
int partials_begin(NbodyHandle* h) {
    if (!(h->cross_on && h->tail_pending) || h->cross.parts.n + h->cross.n_recv == 0) return NBODY_OK;
    if (!h->comm_ready) return fail(h, NBODY_ERR_COMM, "world_size > 1 but nbody_comm_init has not been called");
    const nbody::CrossPlan& c = h->cross;
    const size_t S = h->sym_plan.plane_stride;
    const size_t bytes = size_t(h->sh.seg_cap) * sizeof(float4);
    HIP_TRY(h, hipEventRecord(h->ev_partials_ready, h->stream));
    HIP_TRY(h, hipStreamWaitEvent(h->comm_stream, h->ev_partials_ready, 0));
    TP_TRY(h, h->tp->group_begin());
    for (int i = 0; i < c.parts.n; ++i)
        TP_TRY(h, h->tp->send(h->d_send + size_t(i) * S, bytes, c.parts.seg[i], h->comm_stream));
    for (int i = 0; i < c.n_recv; ++i)
        TP_TRY(h, h->tp->recv(h->d_planes + size_t(h->recv_plane0 + i) * S, bytes, c.recv_from[i], h->comm_stream));
    TP_TRY(h, h->tp->group_end());
    HIP_TRY(h, hipEventRecord(h->ev_partials_done, h->comm_stream));
    h->partials_in_flight = true;
    return NBODY_OK;
}

int partials_wait(NbodyHandle* h) {
    if (!h->partials_in_flight) return NBODY_OK;
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_partials_done, 0));
    h->partials_in_flight = false;
    return NBODY_OK;
}

// everything of the force pass that needs no other GPU's partial sums
int forces_begin(NbodyHandle* h) {
    return h->cfg.method == NBODY_BARNES_HUT ? bh_forces(h) : bf_forces(h);
}

// the plane reduction (with the kick + half drift when a step asked for it)
int forces_finish(NbodyHandle* h) {
    if (!h->tail_pending) return NBODY_OK;
    h->tail_pending = false;
    const float eps2 = h->g_soft * h->g_soft;
    nbody::launch_bf_sym_tail(h->stream, h->sh, h->sym_plan, h->d_planes, int(h->n_local), h->g, eps2,
                              h->kick_pending ? &h->kick_dt : nullptr);
    h->kick_pending = false;
    HIP_TRY(h, hipGetLastError());
    return NBODY_OK;
}

int forces(NbodyHandle* h) {
    int rc = forces_begin(h);
    if (rc) return rc;
    rc = partials_begin(h);
    if (rc) return rc;
    rc = partials_wait(h);
    if (rc) return rc;
    return forces_finish(h);
}

int step_begin(NbodyHandle* h, float dt) {
    if (!h->bounds_set) return fail(h, NBODY_ERR_INVALID, "nbody_set_bounds has not been called");
    nbody::launch_drift_half(h->stream, h->sh, int(h->n_local), dt, h->bnd);  // integrate_pre_force
    nbody::launch_compact(h->stream, h->sh, int(h->n_local));                 // retain
    h->count_dirty = true;
    HIP_TRY(h, hipGetLastError());
    return NBODY_OK;
}

// update_forces up to (not including) whatever needs other GPUs' partial sums
int step_forces(NbodyHandle* h, float dt) {
    h->kick_pending = true;   // a force pass that ends in a plane reduction applies the kick itself
    h->kick_dt = dt;
    int rc = forces_begin(h);                                                  // update_forces
    if (rc) { h->kick_pending = false; h->tail_pending = false; }
    return rc;
}

int step_finish(NbodyHandle* h, float dt) {
    int rc = forces_finish(h);
    if (rc) { h->kick_pending = false; return rc; }
    if (h->kick_pending) nbody::launch_kick_drift(h->stream, h->sh, int(h->n_local), dt);  // integrate_after_force
    h->kick_pending = false;
    HIP_TRY(h, hipGetLastError());
    if (h->ev_pending.size() >= 4096) {  // profiling left on over a long run: fold the timings in now and then
        rc = drain_events(h);
        if (rc) return rc;
    }
    h->elapsed += dt;                                                          // elapsed += dt
    h->stats.steps += 1;
    return NBODY_OK;
}

int step_end(NbodyHandle* h, float dt) {
    int rc = step_forces(h, dt);
    if (rc) return rc;
    rc = partials_begin(h);
    if (rc) return rc;
    rc = partials_wait(h);
    if (rc) return rc;
    return step_finish(h, dt);
}

int step_impl(NbodyHandle* h, float dt) {
    const float elapsed_before = h->elapsed;
    int rc = step_begin(h, dt);
    if (rc) return rc;
    rc = exchange_begin(h);   // the force pass waits for it where it first needs remote bodies
    if (rc) return rc;
    rc = step_end(h, dt);
    if (rc) return rc;
    if (h->last_step_async) {   // nothing was read back: remember the step until the device's progress is confirmed
        h->pending.push_back(NbodyHandle::PendingStep{dt, elapsed_before});
        if (h->pending.size() >= 4096) rc = resolve_async(h);
    }
    return rc;
}

void free_all(NbodyHandle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
    h->tp.reset();   // (leaves the world: the ipc transport waits, bounded, until its peers are done with its window)
    if (h->ev_drifted) (void)hipEventDestroy(h->ev_drifted);
    if (h->ev_gathered) (void)hipEventDestroy(h->ev_gathered);
    if (h->ev_partials_ready) (void)hipEventDestroy(h->ev_partials_ready);
    if (h->ev_partials_done) (void)hipEventDestroy(h->ev_partials_done);
    if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
    for (auto& ev : h->ev_pending) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto& ev : h->ev_free) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    h->tree.clear();
    nbody64::destroy(h);
    nbody::let::destroy(h);
    void* dev[] = {h->sh.pos_all, h->sh.vel, h->sh.acc, h->sh.seg_count, h->sh.escaped, h->sh.keep, h->sh.tile_state, h->sh.epoch, h->sh.inter, h->d_poison, h->d_aos,
                   h->d_nodes, h->d_order, h->d_split, h->d_walk_planes, h->d_walk, h->d_unified, h->d_hot, h->d_hot_info, h->d_bfs, h->d_bfs_ws, h->d_tree_ws, h->d_tree_cat, h->d_nested_stack, h->d_tree_info, h->d_counters, h->d_energy, h->d_sym_bounds, h->d_planes, h->d_cross_slices, h->d_xplanes, h->d_send};
    for (void* p : dev) if (p) (void)hipFree(p);
    void* host[] = {h->h_aos, h->h_pos, h->h_counts, h->h_counters, h->h_split, h->h_tree_info, h->h_hot_info, h->h_poison};
    for (void* p : host) if (p) (void)hipHostFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int create_impl(const NbodyConfig* cfg, NbodyHandle** out) {
    if (!cfg || !out) return fail(nullptr, NBODY_ERR_INVALID, "null argument");
    // ABI versions <= 2 end before shard_mode (48 bytes): such a caller gets index-block shards
    constexpr uint32_t kOldConfigSize = 48;
    NbodyConfig full{};
    if (cfg->struct_size == kOldConfigSize) { std::memcpy(&full, cfg, kOldConfigSize); full.struct_size = sizeof(NbodyConfig); cfg = &full; }
    if (cfg->struct_size != sizeof(NbodyConfig)) return fail(nullptr, NBODY_ERR_INVALID, "NbodyConfig.struct_size mismatch");
    if (cfg->shard_mode != NBODY_SHARD_INDEX && cfg->shard_mode != NBODY_SHARD_SPATIAL) return fail(nullptr, NBODY_ERR_INVALID, "unknown shard_mode");
    if (cfg->shard_mode == NBODY_SHARD_SPATIAL && (cfg->method != NBODY_BARNES_HUT || cfg->math_mode != NBODY_MATH_FAST || cfg->dtype != NBODY_F32))
        return fail(nullptr, NBODY_ERR_INVALID, "NBODY_SHARD_SPATIAL is the fast-math f32 Barnes-Hut path (halo exchange over the device-built tree)");
    if (cfg->method != NBODY_BRUTE_FORCE && cfg->method != NBODY_BARNES_HUT) return fail(nullptr, NBODY_ERR_INVALID, "unknown method");
    if (cfg->math_mode != NBODY_MATH_STRICT && cfg->math_mode != NBODY_MATH_FAST) return fail(nullptr, NBODY_ERR_INVALID, "unknown math_mode");
    if (cfg->leaf_mode != NBODY_LEAF_REFERENCE && cfg->leaf_mode != NBODY_LEAF_DIRECT) return fail(nullptr, NBODY_ERR_INVALID, "unknown leaf_mode");
    if (cfg->tree_build != NBODY_TREE_HOST && cfg->tree_build != NBODY_TREE_DEVICE && cfg->tree_build != NBODY_TREE_AUTO)
        return fail(nullptr, NBODY_ERR_INVALID, "unknown tree_build");
    if (cfg->world_size < 1 || cfg->rank < 0 || cfg->rank >= cfg->world_size) return fail(nullptr, NBODY_ERR_INVALID, "bad rank/world_size");
    if (cfg->capacity == 0 || cfg->capacity > (1ull << 30)) return fail(nullptr, NBODY_ERR_INVALID, "capacity must be in [1, 2^30]");
    if (cfg->dtype != NBODY_F32 && cfg->dtype != NBODY_F64) return fail(nullptr, NBODY_ERR_INVALID, "unknown dtype");
    if (cfg->dtype == NBODY_F64 && cfg->world_size != 1 && cfg->shard_mode != NBODY_SHARD_INDEX)
        return fail(nullptr, NBODY_ERR_INVALID, "f64 worlds are sharded by index blocks (NBODY_SHARD_INDEX)");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, NBODY_ERR_NO_DEVICE, "no HIP device (this library has no CPU fallback)");
    int dev = cfg->device;
    if (dev < 0) {
        const char* lr = std::getenv("LOCAL_RANK");
        dev = lr ? std::atoi(lr) % ndev : 0;
    }
    if (dev >= ndev) return fail(nullptr, NBODY_ERR_INVALID, "device ordinal out of range");

    NbodyHandle* h = new NbodyHandle();
    h->cfg = *cfg;
    if (h->cfg.tree_build == NBODY_TREE_AUTO)   // the bit-exact path keeps the reference's (host) build
        h->cfg.tree_build = cfg->math_mode == NBODY_MATH_FAST ? NBODY_TREE_DEVICE : NBODY_TREE_HOST;
    if (h->cfg.dtype == NBODY_F64) {
        // F = f64: brute force always runs the strict kernel; Barnes-Hut strict = the reference's nested sums on the host-built
        // tree (bit-exact) unless the device build is asked for, fast = one running sum per lane over a split node range, on
        // the device-built tree unless the host build is asked for (AUTO: as for f32)
        if (cfg->method == NBODY_BRUTE_FORCE) h->cfg.math_mode = NBODY_MATH_STRICT;
        if (cfg->tree_build == NBODY_TREE_AUTO) h->cfg.tree_build = h->cfg.math_mode == NBODY_MATH_FAST ? NBODY_TREE_DEVICE : NBODY_TREE_HOST;
        if (cfg->world_size > 1 && h->cfg.math_mode != NBODY_MATH_FAST) h->cfg.tree_build = NBODY_TREE_HOST;   // (a sharded f64 world in strict math builds the replicated tree on the host; fast math: on the device, from the gathered positions)
    }
    if (h->cfg.shard_mode == NBODY_SHARD_SPATIAL) h->cfg.tree_build = NBODY_TREE_DEVICE;
    cfg = &h->cfg;
    h->device = dev;
    *out = nullptr;
    auto bail = [&](int rc) { g_create_err = h->err; free_all(h); return rc; };
#define CREATE_TRY(expr)                                                                                  \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) { h->err = std::string(#expr) + ": " + hipGetErrorString(e_); return bail(NBODY_ERR_HIP); } \
    } while (0)
    CREATE_TRY(hipSetDevice(dev));
    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    if (cfg->dtype == NBODY_F64) {   // the state of an f64 handle lives in nbody64::State
        CREATE_TRY(hipHostMalloc(&h->h_poison, 8 * sizeof(int), hipHostMallocDefault));
        if (cfg->method == NBODY_BARNES_HUT) {
            int threads = cfg->host_threads > 0 ? cfg->host_threads : std::min(16, std::max(1, int(std::thread::hardware_concurrency()) - 2));
            h->pool.reset(new nbody::WorkerPool(std::max(1, threads)));
            CREATE_TRY(hipMalloc(&h->d_counters, 2 * NBODY_WALK_COUNTER_SLOTS * sizeof(unsigned long long)));
            CREATE_TRY(hipMemsetAsync(h->d_counters, 0, 2 * NBODY_WALK_COUNTER_SLOTS * sizeof(unsigned long long), h->stream));
            CREATE_TRY(hipHostMalloc(&h->h_counters, 2 * NBODY_WALK_COUNTER_SLOTS * sizeof(unsigned long long), hipHostMallocDefault));
        }
        h->sh.n_seg = cfg->world_size; h->sh.my_seg = cfg->rank; h->sh.seg_cap = int((cfg->capacity + cfg->world_size - 1) / cfg->world_size);
        h->seg_count_host.assign(size_t(cfg->world_size), 0);
        int rc64 = nbody64::create(h);
        if (rc64) return bail(rc64);
        *out = h;
        return NBODY_OK;
    }
    Shard& sh = h->sh;
    sh.n_seg = cfg->world_size;
    sh.my_seg = cfg->rank;
    sh.seg_cap = int((cfg->capacity + cfg->world_size - 1) / cfg->world_size);
    const bool spatial = cfg->shard_mode == NBODY_SHARD_SPATIAL;
    if (spatial) {   // a spatial handle holds its own bodies only (no gathered positions): one segment, with room for immigrants
        sh.n_seg = 1;
        sh.my_seg = 0;
        // (four times the even share: the bounds equalise the ranks' WORK, and a rank of cheap bodies -- a sparse halo --
        // owns more than the average; NBODY_ERR_CAPACITY beyond that)
        if (cfg->world_size > 1) sh.seg_cap = int(std::min<uint64_t>(cfg->capacity, 4 * uint64_t(sh.seg_cap) + 64));
    }
    const size_t cap = size_t(sh.seg_cap);
    CREATE_TRY(hipMalloc(&sh.pos_all, size_t(sh.n_seg) * cap * sizeof(float4)));
    CREATE_TRY(hipMalloc(&sh.vel, cap * sizeof(float4)));
    CREATE_TRY(hipMalloc(&sh.acc, cap * sizeof(float4)));
    CREATE_TRY(hipMalloc(&sh.seg_count, sizeof(int) * sh.n_seg));
    CREATE_TRY(hipMalloc(&sh.escaped, sizeof(int)));
    CREATE_TRY(hipMalloc(&sh.keep, cap));
    CREATE_TRY(hipMemsetAsync(sh.pos_all, 0, size_t(sh.n_seg) * cap * sizeof(float4), h->stream));
    CREATE_TRY(hipMemsetAsync(sh.vel, 0, cap * sizeof(float4), h->stream));
    CREATE_TRY(hipMemsetAsync(sh.acc, 0, cap * sizeof(float4), h->stream));
    CREATE_TRY(hipMemsetAsync(sh.seg_count, 0, sizeof(int) * sh.n_seg, h->stream));
    CREATE_TRY(hipMemsetAsync(sh.escaped, 0, sizeof(int), h->stream));
    CREATE_TRY(hipMemsetAsync(sh.keep, 1, cap, h->stream));
    {
        const size_t tiles = (cap + 1023) / 1024 + 1;
        CREATE_TRY(hipMalloc(&sh.tile_state, tiles * sizeof(unsigned long long)));
        CREATE_TRY(hipMemsetAsync(sh.tile_state, 0, tiles * sizeof(unsigned long long), h->stream));
        CREATE_TRY(hipMalloc(&sh.epoch, sizeof(int)));
        CREATE_TRY(hipMemsetAsync(sh.epoch, 0, sizeof(int), h->stream));
        CREATE_TRY(hipMemsetAsync(sh.epoch, 1, 1, h->stream));   // epoch = 1: the zeroed status words belong to no launch
        CREATE_TRY(hipMalloc(&sh.inter, sizeof(unsigned long long)));
        CREATE_TRY(hipMemsetAsync(sh.inter, 0, sizeof(unsigned long long), h->stream));
        CREATE_TRY(hipMalloc(&h->d_poison, 2 * sizeof(int)));
        CREATE_TRY(hipMemsetAsync(h->d_poison, 0, 2 * sizeof(int), h->stream));
        CREATE_TRY(hipHostMalloc(&h->h_poison, 8 * sizeof(int), hipHostMallocDefault));
    }
    CREATE_TRY(hipHostMalloc(&h->h_counts, sizeof(int) * sh.n_seg, hipHostMallocDefault));
    h->seg_count_host.assign(sh.n_seg, 0);
    if (cfg->method == NBODY_BARNES_HUT) {
        // default: the cores this process may run on, at most 16 (a GPU's share of the host; more
        // threads than top-level subtrees only add wake-up latency)
        int threads = cfg->host_threads > 0 ? cfg->host_threads : std::min(16, std::max(1, int(std::thread::hardware_concurrency()) - 2));
        if (threads < 1) threads = 1;
        h->pool.reset(new nbody::WorkerPool(threads));
        h->tree.alloc = pinned_alloc;
        h->tree.release = pinned_free;
        CREATE_TRY(hipHostMalloc(&h->h_pos, size_t(sh.n_seg) * cap * sizeof(float4), hipHostMallocDefault));
        CREATE_TRY(hipMalloc(&h->d_counters, 2 * NBODY_WALK_COUNTER_SLOTS * sizeof(unsigned long long)));
        CREATE_TRY(hipMemsetAsync(h->d_counters, 0, 2 * NBODY_WALK_COUNTER_SLOTS * sizeof(unsigned long long), h->stream));
        CREATE_TRY(hipHostMalloc(&h->h_counters, 2 * NBODY_WALK_COUNTER_SLOTS * sizeof(unsigned long long), hipHostMallocDefault));
    }
    CREATE_TRY(hipStreamSynchronize(h->stream));
#undef CREATE_TRY
    {
        const char* v = std::getenv("NBODY_BH_ASYNC");
        h->async_bh = cfg->method == NBODY_BARNES_HUT && cfg->tree_build == NBODY_TREE_DEVICE && cfg->world_size == 1 &&
                      !spatial && !(v && v[0] == '0');
        if (h->async_bh) sh.poison = h->d_poison;
    }
    if (spatial) {
        int rc_let = nbody::let::create(h);
        if (rc_let) return bail(rc_let);
    }
    {   // the documented environment switches set this handle's knobs (nbody_set_tuning changes them later)
        nbody::Tuning& t = h->tune;
        const struct { const char* env; int* knob; } table[] = {
            {"NBODY_BF_VARIANT", &t.bf_fast_variant}, {"NBODY_CROSS_SYM", &t.cross_sym}, {"NBODY_SYM_PACKED", &t.sym_packed},
            {"NBODY_BH_SPLIT", &t.bh_walk_split}, {"NBODY_SYM_WPB", &t.sym_wpb}, {"NBODY_SYM_IPT", &t.sym_ipt}, {"NBODY_BH_DUO", &t.bh_walk_duo}, {"NBODY_BH_XCD", &t.bh_walk_xcd},
#ifdef NBODY_TUNING
            {"NBODY_BH_VARIANT", &t.bh_walk_variant}, {"NBODY_BH_HOT", &t.bh_hot_cap}, {"NBODY_BH_LDS_BLOCK", &t.bh_walk_lds_block},
#endif
        };
        for (const auto& e : table)
            if (const char* v = std::getenv(e.env)) *e.knob = std::atoi(v);
    }
    *out = h;
    return NBODY_OK;
}

}  // namespace

// =============================================================================== C entry points
extern "C" {

int nbody_abi_version(void) { return NBODY_ABI_VERSION; }

int nbody_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* nbody_last_error(const NbodyHandle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int nbody_create(const NbodyConfig* cfg, NbodyHandle** out) { return create_impl(cfg, out); }

void nbody_destroy(NbodyHandle* h) { free_all(h); }

int nbody_clone(const NbodyHandle* src, NbodyHandle** out) {
    if (!src || !out) return fail(nullptr, NBODY_ERR_INVALID, "null argument");
    NbodyHandle* s = const_cast<NbodyHandle*>(src);
    int rc = use_device(s);
    if (rc) return rc;
    rc = resolve_async(s);
    if (rc) return rc;
    if (!s->f64) rc = sync_count(s);
    if (rc) return rc;
    NbodyHandle* h = nullptr;
    rc = create_impl(&src->cfg, &h);
    if (rc) return rc;
    if (src->f64) {
        rc = nbody64::clone_state(s, h);
        if (rc) { g_create_err = h->err; free_all(h); return rc; }
        *out = h;
        return NBODY_OK;
    }
    const Shard& a = src->sh;
    const size_t cap = size_t(a.seg_cap);
    hipError_t e = hipStreamSynchronize(s->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h->sh.pos_all, a.pos_all, size_t(a.n_seg) * cap * sizeof(float4), hipMemcpyDeviceToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h->sh.vel, a.vel, cap * sizeof(float4), hipMemcpyDeviceToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h->sh.acc, a.acc, cap * sizeof(float4), hipMemcpyDeviceToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h->sh.seg_count, a.seg_count, sizeof(int) * a.n_seg, hipMemcpyDeviceToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        g_create_err = std::string("clone copy: ") + hipGetErrorString(e);
        free_all(h);
        return NBODY_ERR_HIP;
    }
    h->g = src->g; h->g_soft = src->g_soft; h->dt = src->dt; h->theta2 = src->theta2;
    h->tune = src->tune;
    std::memcpy(h->center, src->center, sizeof(h->center));
    h->width = src->width; h->bnd = src->bnd; h->bounds_set = src->bounds_set;
    h->elapsed = src->elapsed;
    h->n_local = src->n_local;
    h->seg_count_host = src->seg_count_host;
    h->first_global = src->first_global; h->n_at_upload = src->n_at_upload;
    if (src->let) {   // spatial shards: + the bodies' ids and the ownership bounds
        rc = nbody::let::clone_state(s, h);
        if (rc) { g_create_err = h->err; free_all(h); return rc; }
    }
    // like the reference's BH clone (barnes_hut.rs:113-135) the tree is not carried over; neither
    // are the communicator (call nbody_comm_init on the clone) and the statistics
    *out = h;
    return NBODY_OK;
}

int nbody_upload(NbodyHandle* h, const void* aos, size_t n, size_t stride) {
    if (!h || (!aos && n)) return fail(h, NBODY_ERR_INVALID, "null argument");
    if (stride < 40 || stride % 4) return fail(h, NBODY_ERR_INVALID, "stride must be a multiple of 4 and >= 40 bytes");
    if (n > h->cfg.capacity) return fail(h, NBODY_ERR_CAPACITY, "more bodies than NbodyConfig.capacity");
    int rc = use_device(h);
    if (rc) return rc;
    if (h->f64) return (n && !aos) ? NBODY_ERR_INVALID : nbody64::upload(h, aos, n, stride);
    if (h->let) return nbody::let::upload(h, aos, n, stride);
    rc = resolve_async(h);
    if (rc) return rc;
    Shard& sh = h->sh;
    const size_t G = size_t(sh.n_seg);
    const size_t blk = (n + G - 1) / G;  // contiguous index blocks keep the ascending-partner order
    rc = ensure_aos(h, n);
    if (rc) return rc;
    const char* src = static_cast<const char*>(aos);
    for (size_t k = 0; k < n; ++k) std::memcpy(h->h_aos + 10 * k, src + k * stride, 40);
    if (n) HIP_TRY(h, hipMemcpyAsync(h->d_aos, h->h_aos, n * 40, hipMemcpyHostToDevice, h->stream));
    for (size_t s = 0; s < G; ++s) {
        size_t lo = std::min(n, s * blk), hi = std::min(n, lo + blk);
        h->seg_count_host[s] = int(hi - lo);
        bool own = int(s) == sh.my_seg;
        nbody::launch_aos_to_soa(h->stream, h->d_aos + 10 * lo, 10, int(hi - lo), sh.pos_all + s * size_t(sh.seg_cap),
                                 own ? sh.vel : nullptr, own ? sh.acc : nullptr);
        if (own) { h->first_global = lo; h->n_at_upload = hi - lo; h->n_local = hi - lo; }
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemsetAsync(sh.escaped, 0, sizeof(int), h->stream));
    h->count_dirty = false;
    return push_counts(h);
}

int nbody_download(NbodyHandle* h, void* aos, size_t cap, size_t stride, size_t* n_out) {
    if (!h) return NBODY_ERR_INVALID;
    if (stride < 40 || stride % 4) return fail(h, NBODY_ERR_INVALID, "stride must be a multiple of 4 and >= 40 bytes");
    int rc = use_device(h);
    if (rc) return rc;
    if (h->f64) return nbody64::download(h, aos, cap, stride, n_out);
    rc = resolve_async(h);
    if (rc) return rc;
    rc = sync_count(h);
    if (rc) return rc;
    const size_t n = h->n_local;
    if (n_out) *n_out = n;
    if (n > cap) return fail(h, NBODY_ERR_CAPACITY, "download buffer too small");
    if (n == 0) return NBODY_OK;
    if (!aos) return fail(h, NBODY_ERR_INVALID, "null buffer");
    rc = ensure_aos(h, n);
    if (rc) return rc;
    nbody::launch_soa_to_aos(h->stream, h->d_aos, 10, int(n), h->sh.own_pos(), h->sh.vel, h->sh.acc);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(h->h_aos, h->d_aos, n * 40, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    rc = comm_check(h);   // (bodies of a run whose exchange broke down are not handed out as results)
    if (rc) return rc;
    char* dst = static_cast<char*>(aos);
    for (size_t k = 0; k < n; ++k) std::memcpy(dst + k * stride, h->h_aos + 10 * k, 40);
    return NBODY_OK;
}

int nbody_count(NbodyHandle* h, size_t* n_out) {
    if (!h || !n_out) return NBODY_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (h->f64) return nbody64::count(h, n_out);
    rc = resolve_async(h);
    if (rc) return rc;
    rc = sync_count(h);
    if (rc) return rc;
    *n_out = h->n_local;
    return NBODY_OK;
}

int nbody_count_global(NbodyHandle* h, size_t* n_out) {
    if (!h || !n_out) return NBODY_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (h->f64) return nbody64::count_global(h, n_out);
    if (h->let) return nbody::let::count_global(h, n_out);
    rc = resolve_async(h);
    if (rc) return rc;
    rc = sync_count(h);
    if (rc) return rc;
    *n_out = total_upper(h);
    return NBODY_OK;
}

int nbody_add_point(NbodyHandle* h, const void* particle) {
    if (!h || !particle) return NBODY_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (h->f64) return nbody64::add_point(h, particle);
    if (h->let) return nbody::let::add_point(h, particle);
    if (h->sh.n_seg != 1) return sharded_add_point(h, particle);
    rc = resolve_async(h);
    if (rc) return rc;
    rc = sync_count(h);
    if (rc) return rc;
    if (h->n_local >= size_t(h->sh.seg_cap)) return fail(h, NBODY_ERR_CAPACITY, "capacity exhausted");
    const float* p = static_cast<const float*>(particle);
    float4 rec[3] = {make_float4(p[0], p[1], p[2], p[9]), make_float4(p[3], p[4], p[5], 0.f), make_float4(p[6], p[7], p[8], 0.f)};
    const size_t k = h->n_local;
    HIP_TRY(h, hipMemcpyAsync(h->sh.pos_all + k, &rec[0], sizeof(float4), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->sh.vel + k, &rec[1], sizeof(float4), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->sh.acc + k, &rec[2], sizeof(float4), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->n_local = k + 1;
    h->seg_count_host[0] = int(h->n_local);
    return push_counts(h);
}

int nbody_remove_point(NbodyHandle* h, size_t index) {
    if (!h) return NBODY_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (h->f64) return nbody64::remove_point(h, index);
    if (h->let) return nbody::let::remove_point(h, index);
    if (h->sh.n_seg != 1) return sharded_remove_point(h, index);
    rc = resolve_async(h);
    if (rc) return rc;
    rc = sync_count(h);
    if (rc) return rc;
    if (index >= h->n_local) return fail(h, NBODY_ERR_INVALID, "swap_remove index out of range");  // Vec::swap_remove panics
    const size_t last = h->n_local - 1;
    if (index != last) {
        HIP_TRY(h, hipMemcpyAsync(h->sh.pos_all + index, h->sh.pos_all + last, sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->sh.vel + index, h->sh.vel + last, sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->sh.acc + index, h->sh.acc + last, sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
    }
    h->n_local = last;
    h->seg_count_host[0] = int(last);
    return push_counts(h);
}

int nbody_set_settings(NbodyHandle* h, float g, float g_soft, float dt, float theta2) {
    if (!h) return NBODY_ERR_INVALID;
    if (h->f64) return nbody64::set_settings(h, double(g), double(g_soft), double(dt), double(theta2));
    h->g = g; h->g_soft = g_soft; h->dt = dt; h->theta2 = theta2;
    return NBODY_OK;
}

int nbody_get_settings(const NbodyHandle* h, float* g, float* g_soft, float* dt, float* theta2) {
    if (!h) return NBODY_ERR_INVALID;
    if (h->f64) {
        double a, b, c, d;
        nbody64::get_settings(h, &a, &b, &c, &d);
        if (g) *g = float(a);
        if (g_soft) *g_soft = float(b);
        if (dt) *dt = float(c);
        if (theta2) *theta2 = float(d);
        return NBODY_OK;
    }
    if (g) *g = h->g;
    if (g_soft) *g_soft = h->g_soft;
    if (dt) *dt = h->dt;
    if (theta2) *theta2 = h->theta2;
    return NBODY_OK;
}

int nbody_set_settings_f64(NbodyHandle* h, double g, double g_soft, double dt, double theta2) {
    if (!h) return NBODY_ERR_INVALID;
    if (h->f64) return nbody64::set_settings(h, g, g_soft, dt, theta2);
    return nbody_set_settings(h, float(g), float(g_soft), float(dt), float(theta2));
}

int nbody_get_settings_f64(const NbodyHandle* h, double* g, double* g_soft, double* dt, double* theta2) {
    if (!h) return NBODY_ERR_INVALID;
    if (h->f64) return nbody64::get_settings(h, g, g_soft, dt, theta2);
    if (g) *g = double(h->g);
    if (g_soft) *g_soft = double(h->g_soft);
    if (dt) *dt = double(h->dt);
    if (theta2) *theta2 = double(h->theta2);
    return NBODY_OK;
}

int nbody_set_bounds_f64(NbodyHandle* h, const double center[3], double width) {
    if (!h || !center) return NBODY_ERR_INVALID;
    if (h->f64) return nbody64::set_bounds(h, center, width);
    const float c[3] = {float(center[0]), float(center[1]), float(center[2])};
    return nbody_set_bounds(h, c, float(width));
}

int nbody_set_bounds(NbodyHandle* h, const float center[3], float width) {
    if (!h || !center) return NBODY_ERR_INVALID;
    if (h->f64) {
        const double c[3] = {double(center[0]), double(center[1]), double(center[2])};
        return nbody64::set_bounds(h, c, double(width));
    }
    std::memcpy(h->center, center, sizeof(h->center));
    h->width = width;
    compute_bounds(h);
    h->bounds_set = true;
    return NBODY_OK;
}

int nbody_init(NbodyHandle* h) {
    if (!h) return NBODY_ERR_INVALID;
    if (h->f64) return nbody64::init(h);
    h->elapsed = 0.f;  // brute_force.rs:49; the reference's BH init also builds a tree that the
                       // first update_forces rebuilds before any use (barnes_hut.rs:232-235, 251-254)
    return NBODY_OK;
}

int nbody_step_by(NbodyHandle* h, float dt) {
    if (!h) return NBODY_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (h->f64) return nbody64::step_by(h, double(dt));
    if (h->let) return nbody::let::step(h, dt);
    return step_impl(h, dt);
}

int nbody_step_by_f64(NbodyHandle* h, double dt) {
    if (!h) return NBODY_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (h->f64) return nbody64::step_by(h, dt);
    if (h->let) return nbody::let::step(h, float(dt));
    return step_impl(h, float(dt));
}

int nbody_steps(NbodyHandle* h, int k) {
    if (!h || k < 0) return NBODY_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (h->f64) return nbody64::steps(h, k);
    if (h->let) {
        for (int i = 0; i < k; ++i) { rc = nbody::let::step(h, h->dt); if (rc) return rc; }
        return NBODY_OK;
    }
    for (int i = 0; i < k; ++i) {
        rc = step_impl(h, h->dt);  // Simulation::step, shared.rs:86-88
        if (rc) return rc;
    }
    return NBODY_OK;
}

int nbody_update_forces(NbodyHandle* h) {
    if (!h) return NBODY_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (h->f64) return nbody64::update_forces(h);
    if (h->let) return h->bounds_set ? nbody::let::update_forces(h) : fail(h, NBODY_ERR_INVALID, "nbody_set_bounds has not been called");
    if (h->cfg.method == NBODY_BARNES_HUT && !h->bounds_set) return fail(h, NBODY_ERR_INVALID, "nbody_set_bounds has not been called");
    rc = resolve_async(h);
    if (rc) return rc;
    rc = exchange_begin(h);
    if (rc) return rc;
    rc = forces(h);
    if (rc || !h->last_step_async) return rc;
    h->last_step_async = false;
    rc = resolve_async(h);              // (a force pass outside a step is confirmed at once)
    if (rc || !h->host_tree_once) return rc;
    return forces(h);                   // its build needed the host: once more, on the host-built tree
}

int nbody_elapsed(const NbodyHandle* h, float* out) {
    if (!h || !out) return NBODY_ERR_INVALID;
    *out = h->f64 ? float(nbody64::elapsed(h)) : h->elapsed;
    return NBODY_OK;
}

int nbody_elapsed_f64(const NbodyHandle* h, double* out) {
    if (!h || !out) return NBODY_ERR_INVALID;
    *out = h->f64 ? nbody64::elapsed(h) : double(h->elapsed);
    return NBODY_OK;
}

int nbody_sync(NbodyHandle* h) {
    if (!h) return NBODY_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    rc = resolve_async(h);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return comm_check(h);
}

int nbody_set_profiling(NbodyHandle* h, int on) {
    if (!h) return NBODY_ERR_INVALID;
    h->profiling = on != 0;
    h->profile_every = on > 1 ? on : 1;
    h->profile_tick = 0;
    return NBODY_OK;
}

int nbody_stats(NbodyHandle* h, NbodyStats* out) {
    if (!h || !out) return NBODY_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    rc = resolve_async(h);
    if (rc) return rc;
    rc = drain_events(h);
    if (rc) return rc;
    if (h->f64) return nbody64::stats(h, out);
    if (h->cfg.method == NBODY_BRUTE_FORCE) {
        unsigned long long* hv = reinterpret_cast<unsigned long long*>(h->h_poison + 4);   // (pinned scratch)
        HIP_TRY(h, hipMemcpyAsync(hv, h->sh.inter, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        h->stats.interactions = *hv;
    }
    if (h->d_counters) {
        HIP_TRY(h, hipMemcpyAsync(h->h_counters, h->d_counters, 2 * NBODY_WALK_COUNTER_SLOTS * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        unsigned long long acc_sum = 0, vis_sum = 0;
        for (unsigned k = 0; k < NBODY_WALK_COUNTER_SLOTS; ++k) { acc_sum += h->h_counters[2 * k]; vis_sum += h->h_counters[2 * k + 1]; }
        h->stats.interactions = acc_sum;
        h->stats.force_kernel_interactions = acc_sum;  // the walk kernel evaluates all of them
        h->stats.node_visits = vis_sum;
    }
    *out = h->stats;
    return NBODY_OK;
}

int nbody_reset_stats(NbodyHandle* h) {
    if (!h) return NBODY_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    rc = resolve_async(h);
    if (rc) return rc;
    rc = drain_events(h);
    if (rc) return rc;
    uint64_t nodes = h->stats.tree_nodes;
    h->stats = NbodyStats{};
    h->stats.tree_nodes = nodes;
    if (h->let) { rc = nbody::let::reset_stats(h); if (rc) return rc; }
    if (h->f64) { rc = nbody64::reset_stats(h); if (rc) return rc; }
    else HIP_TRY(h, hipMemsetAsync(h->sh.inter, 0, sizeof(unsigned long long), h->stream));
    if (h->d_counters) {
        HIP_TRY(h, hipMemsetAsync(h->d_counters, 0, 2 * NBODY_WALK_COUNTER_SLOTS * sizeof(unsigned long long), h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return NBODY_OK;
}

int nbody_energy(NbodyHandle* h, double* kinetic, double* potential) {
    if (!h) return NBODY_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (h->f64) return nbody64::energy(h, kinetic, potential);
    if (h->let && h->cfg.world_size > 1)   // (a spatial rank sees its own bodies only: the pair sum over them is not the world's energy)
        return fail(h, NBODY_ERR_INVALID, "nbody_energy is not supported on NBODY_SHARD_SPATIAL handles of a world of more than one rank");
    rc = resolve_async(h);
    if (rc) return rc;
    rc = sync_count(h);
    if (rc) return rc;
    const size_t n = h->n_local;
    const size_t blocks = (n + 255) / 256;
    double ke = 0.0, pe = 0.0;
    if (blocks) {
        if (blocks > h->energy_blocks) {
            if (h->d_energy) (void)hipFree(h->d_energy);
            h->d_energy = nullptr; h->energy_blocks = 0;
            HIP_TRY(h, hipMalloc(&h->d_energy, blocks * 2 * sizeof(double)));
            h->energy_blocks = blocks;
        }
        nbody::launch_energy(h->stream, h->sh, int(n), double(h->g_soft) * double(h->g_soft), h->d_energy);
        HIP_TRY(h, hipGetLastError());
        std::vector<double> part(blocks * 2);
        HIP_TRY(h, hipMemcpyAsync(part.data(), h->d_energy, blocks * 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        for (size_t b = 0; b < blocks; ++b) { ke += part[2 * b]; pe += part[2 * b + 1]; }
    }
    if (kinetic) *kinetic = ke;
    if (potential) *potential = -0.5 * double(h->g) * pe;  // every unordered pair was met twice
    return NBODY_OK;
}

int nbody_tree_export(NbodyHandle* h, float* com_mass, float* width, int32_t* skip, size_t cap, size_t* n_nodes) {
    if (!h) return NBODY_ERR_INVALID;
    if (h->cfg.method != NBODY_BARNES_HUT) return fail(h, NBODY_ERR_INVALID, "not a Barnes-Hut handle");
    if (h->f64) {
        if (com_mass || width) return fail(h, NBODY_ERR_INVALID, "f64 handle: use nbody_tree_export_f64");
        return nbody64::tree_export(h, nullptr, nullptr, skip, cap, n_nodes);
    }
    if (h->let)   // (a spatial rank holds its slice and what it imported, never the whole tree)
        return fail(h, NBODY_ERR_INVALID, "nbody_tree_export is not supported on NBODY_SHARD_SPATIAL handles");
    {
        int rc = use_device(h);
        if (rc) return rc;
        rc = resolve_async(h);
        if (rc) return rc;
    }
    const size_t n = h->tree.n_nodes;
    if (n_nodes) *n_nodes = n;
    if (!com_mass && !width && !skip) return NBODY_OK;
    if (h->tree_on_device) {  // the octree lives on the device only: fetch it
        int rc = use_device(h);
        if (rc) return rc;
        h->tree.reserve(n, 0);
        HIP_TRY(h, hipMemcpyAsync(h->tree.nodes, h->d_nodes, n * sizeof(nbody::NodeRec), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    if (n > cap) return fail(h, NBODY_ERR_CAPACITY, "tree export buffer too small");
    for (size_t i = 0; i < n; ++i) {
        const nbody::NodeRec& r = h->tree.nodes[i];
        if (com_mass) { com_mass[4 * i] = r.a.x; com_mass[4 * i + 1] = r.a.y; com_mass[4 * i + 2] = r.a.z; com_mass[4 * i + 3] = r.a.m; }
        if (width) width[i] = std::sqrt(r.b.w2);  // exact: w2 is the rounded square of the width
        if (skip) skip[i] = r.b.skip;
    }
    return NBODY_OK;
}

int nbody_tree_export_f64(NbodyHandle* h, double* com_mass, double* width, int32_t* skip, size_t cap, size_t* n_nodes) {
    if (!h) return NBODY_ERR_INVALID;
    if (h->cfg.method != NBODY_BARNES_HUT) return fail(h, NBODY_ERR_INVALID, "not a Barnes-Hut handle");
    if (!h->f64) return fail(h, NBODY_ERR_INVALID, "f32 handle: use nbody_tree_export");
    return nbody64::tree_export(h, com_mass, width, skip, cap, n_nodes);
}

// ---- the cells of the last tree, for drawing (the reference's Barnes-Hut Renderable walks every node's bounds,
// barnes_hut.rs:322-343).  The node records carry centre of mass, width and skip link, not the box: it is recovered top
// down with the reference's own recurrences -- a node's orthant in its parent is get_orthant(parent centre, its centre
// of mass) (shared.rs:245-254: all its bodies lie in that orthant, so their centre of mass does), its box create_orthant
// (shared.rs:256-272: centre +- half_width / 2, half_width / 2).
extern "C++" {
namespace {
template <class F>
void cells_from_preorder(const F* com_mass, const int32_t* skip, size_t n, const F root_center[3], F root_width, float* min_max6, int32_t* depth) {
    struct Open { size_t end; F c[3]; F half; int depth; };
    std::vector<Open> stack;
    for (size_t i = 0; i < n; ++i) {
        while (!stack.empty() && stack.back().end <= i) stack.pop_back();
        Open me;
        me.end = size_t(skip[i]);
        if (stack.empty()) {
            for (int k = 0; k < 3; ++k) me.c[k] = root_center[k];
            me.half = root_width * F(0.5);     // Bounds::new
            me.depth = 0;
        } else {
            const Open& p = stack.back();
            me.half = p.half * F(0.5);         // create_orthant
            for (int k = 0; k < 3; ++k) me.c[k] = com_mass[4 * i + k] > p.c[k] ? p.c[k] + me.half : p.c[k] - me.half;
            me.depth = p.depth + 1;
        }
        if (min_max6)
            for (int k = 0; k < 3; ++k) {
                min_max6[6 * i + k] = float(me.c[k] + (-me.half));       // Bounds::min (shared.rs:223-225)
                min_max6[6 * i + 3 + k] = float(me.c[k] + me.half);      // Bounds::max
            }
        if (depth) depth[i] = me.depth;
        if (me.end > i + 1) stack.push_back(me);   // it has children: they follow
    }
}
}  // namespace
}  // extern "C++"

int nbody_tree_export_cells(NbodyHandle* h, float* min_max6, int32_t* depth, size_t cap, size_t* n_nodes) {
    if (!h) return NBODY_ERR_INVALID;
    size_t n = 0;
    if (h->f64) {
        int rc = nbody_tree_export_f64(h, nullptr, nullptr, nullptr, 0, &n);
        if (rc) return rc;
        if (n_nodes) *n_nodes = n;
        if (!min_max6 && !depth) return NBODY_OK;
        if (n > cap) return fail(h, NBODY_ERR_CAPACITY, "tree export buffer too small");
        std::vector<double> cm(4 * n), w(n);
        std::vector<int32_t> sk(n);
        rc = nbody_tree_export_f64(h, cm.data(), w.data(), sk.data(), n, &n);
        if (rc) return rc;
        double c[3], width;
        nbody64::get_bounds(h, c, &width);
        cells_from_preorder<double>(cm.data(), sk.data(), n, c, width, min_max6, depth);
        return NBODY_OK;
    }
    int rc = nbody_tree_export(h, nullptr, nullptr, nullptr, 0, &n);
    if (rc) return rc;
    if (n_nodes) *n_nodes = n;
    if (!min_max6 && !depth) return NBODY_OK;
    if (n > cap) return fail(h, NBODY_ERR_CAPACITY, "tree export buffer too small");
    std::vector<float> cm(4 * n), w(n);
    std::vector<int32_t> sk(n);
    rc = nbody_tree_export(h, cm.data(), w.data(), sk.data(), n, &n);
    if (rc) return rc;
    cells_from_preorder<float>(cm.data(), sk.data(), n, h->center, h->width, min_max6, depth);
    return NBODY_OK;
}

int nbody_comm_unique_id(void* id_bytes) {
    if (!id_bytes) return NBODY_ERR_INVALID;
    // NBODY_TRANSPORT=ipc: ranks that share one device (a one-GPU box rehearsing the multi-rank step); default: RCCL
    const char* v = std::getenv("NBODY_TRANSPORT");
    std::string err;
    int rc = nbody::transport_make_id((v && std::strcmp(v, "ipc") == 0) ? nbody::kTransportIpc : nbody::kTransportRccl, id_bytes, &err);
    return rc ? fail(nullptr, rc, err) : NBODY_OK;
}

int nbody_comm_local_id(void* id_bytes) {
    if (!id_bytes) return NBODY_ERR_INVALID;
    std::string err;
    int rc = nbody::transport_make_id(nbody::kTransportIpc, id_bytes, &err);
    return rc ? fail(nullptr, rc, err) : NBODY_OK;
}

// what every rank of a world must agree on: a rank that chose another exchange scheme than its peers would neither send
// nor expect what the others exchange with it (a hang in RCCL), so disagreement is an error at nbody_comm_init
struct Agreement {
    int32_t abi, method, math_mode, leaf_mode, tree_build, dtype, shard_mode, world, seg_cap;
    int32_t cross_sym, sym_packed, bf_variant, walk_variant, walk_split;
    uint64_t capacity;
};
static const char* const kAgreementFields[] = {"ABI version", "method", "math_mode", "leaf_mode", "tree_build", "dtype", "shard_mode", "world_size",
                                               "shard capacity", "NBODY_CROSS_SYM", "NBODY_SYM_PACKED", "NBODY_BF_VARIANT", "NBODY_BH_VARIANT",
                                               "NBODY_BH_SPLIT"};

int nbody_comm_init(NbodyHandle* h, const void* id_bytes) {
    if (!h || !id_bytes) return NBODY_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    h->comm_ready = false;
    h->tp.reset();
    {
        std::string err;
        int code = NBODY_ERR_COMM;
        h->tp.reset(nbody::transport_create(id_bytes, h->cfg.rank, h->cfg.world_size, h->device, &err, &code));
        if (!h->tp) return fail(h, code, err);
    }
    if (!h->comm_stream) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_drifted, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_gathered, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_partials_ready, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_partials_done, hipEventDisableTiming));
    }
    Agreement mine{NBODY_ABI_VERSION, h->cfg.method, h->cfg.math_mode, h->cfg.leaf_mode, h->cfg.tree_build, h->cfg.dtype, h->cfg.shard_mode,
                   h->cfg.world_size, h->sh.seg_cap, nbody::tuning().cross_sym, nbody::tuning().sym_packed, nbody::tuning().bf_fast_variant, nbody::tuning().bh_walk_variant,
                   nbody::tuning().bh_walk_split, h->cfg.capacity};
    std::vector<Agreement> all(size_t(h->cfg.world_size));
    TP_TRY(h, h->tp->host_all_gather(&mine, all.data(), sizeof(Agreement)));
    for (int r = 0; r < h->cfg.world_size; ++r) {
        const int32_t* a = reinterpret_cast<const int32_t*>(&all[size_t(r)]);
        const int32_t* m = reinterpret_cast<const int32_t*>(&mine);
        for (size_t f = 0; f < sizeof(kAgreementFields) / sizeof(kAgreementFields[0]); ++f)
            if (a[f] != m[f]) {
                h->tp.reset();
                return fail(h, NBODY_ERR_COMM, std::string("nbody_comm_init: rank ") + std::to_string(r) + " and rank " + std::to_string(h->cfg.rank) + " disagree on " +
                                                   kAgreementFields[f] + " (" + std::to_string(a[f]) + " vs " + std::to_string(m[f]) + "): every rank of a world must be created alike");
            }
        if (all[size_t(r)].capacity != mine.capacity) {
            h->tp.reset();
            return fail(h, NBODY_ERR_COMM, "nbody_comm_init: ranks disagree on NbodyConfig.capacity");
        }
    }
    h->comm_ready = true;
    return NBODY_OK;
}

// ---- launch-shape and scheme knobs of one handle (kernels.h struct Tuning)
namespace {
struct Knob { const char* name; int nbody::Tuning::*field; bool tuning_build_only; };
const Knob kKnobs[] = {
    {"cross_sym", &nbody::Tuning::cross_sym, false}, {"sym_packed", &nbody::Tuning::sym_packed, false},
    {"bf_fast_variant", &nbody::Tuning::bf_fast_variant, false}, {"sym_wpb", &nbody::Tuning::sym_wpb, false}, {"sym_ipt", &nbody::Tuning::sym_ipt, false}, {"let_list_div", &nbody::Tuning::let_list_div, false}, {"bh_walk_duo", &nbody::Tuning::bh_walk_duo, false}, {"bh_walk_xcd", &nbody::Tuning::bh_walk_xcd, false},
    {"sym_rounds", &nbody::Tuning::sym_rounds, false}, {"sym_k", &nbody::Tuning::sym_k, false},
    {"sym_min_bodies", &nbody::Tuning::sym_min_bodies, false}, {"sym_reduce_split", &nbody::Tuning::sym_reduce_split, false},
    {"cross_slots", &nbody::Tuning::cross_slots, false}, {"cross_ipt", &nbody::Tuning::cross_ipt, false},
    {"cross_wpb", &nbody::Tuning::cross_wpb, false}, {"bh_walk_split", &nbody::Tuning::bh_walk_split, false},
    {"bh_walk_order", &nbody::Tuning::bh_walk_order, false}, {"bh_reduce_split", &nbody::Tuning::bh_reduce_split, false},
    {"tree_max_tie", &nbody::Tuning::tree_max_tie, false},
    {"bh_walk_variant", &nbody::Tuning::bh_walk_variant, true}, {"bh_walk_lds_block", &nbody::Tuning::bh_walk_lds_block, true},
    {"bh_hot_cap", &nbody::Tuning::bh_hot_cap, true}, {"bh_walk_debug", &nbody::Tuning::bh_walk_debug, true},
    {"sym_debug", &nbody::Tuning::sym_debug, true},
};
}  // namespace

int nbody_set_tuning(NbodyHandle* h, const char* name, int value) {
    if (!h || !name) return NBODY_ERR_INVALID;
    for (const Knob& k : kKnobs)
        if (std::strcmp(k.name, name) == 0) {
#ifndef NBODY_TUNING
            if (k.tuning_build_only && value != nbody::Tuning{}.*(k.field))
                return fail(h, NBODY_ERR_INVALID, std::string("nbody_set_tuning: '") + name + "' selects code only the tuning build carries (make -C nbody-llm_amd/csrc tuning)");
#endif
            if (h->comm_ready && (std::strcmp(name, "cross_sym") == 0 || std::strcmp(name, "sym_packed") == 0 || std::strcmp(name, "bf_fast_variant") == 0))
                return fail(h, NBODY_ERR_INVALID, std::string("nbody_set_tuning: '") + name + "' is part of what the ranks agreed on at nbody_comm_init: set it before");
            h->tune.*(k.field) = value;
            return NBODY_OK;
        }
    return fail(h, NBODY_ERR_INVALID, std::string("nbody_set_tuning: unknown knob '") + name + "'");
}

int nbody_get_tuning(const NbodyHandle* h, const char* name, int* value) {
    if (!h || !name || !value) return NBODY_ERR_INVALID;
    for (const Knob& k : kKnobs)
        if (std::strcmp(k.name, name) == 0) { *value = h->tune.*(k.field); return NBODY_OK; }
    return NBODY_ERR_INVALID;
}

int nbody_is_tuning_build(void) {
#ifdef NBODY_TUNING
    return 1;
#else
    return 0;
#endif
}

int nbody_comm_transport(const NbodyHandle* h, char* out, size_t cap) {
    if (!h || !out || cap == 0) return NBODY_ERR_INVALID;
    std::snprintf(out, cap, "%s", h->tp ? h->tp->name() : "none");
    return NBODY_OK;
}

int nbody_download_ids(NbodyHandle* h, int32_t* ids, size_t cap, size_t* n_out) {
    if (!h) return NBODY_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (!h->let) return fail(h, NBODY_ERR_INVALID, "nbody_download_ids is for NBODY_SHARD_SPATIAL handles (index-block shards: nbody_local_range)");
    return nbody::let::download_ids(h, ids, cap, n_out);
}

int nbody_let_stats(NbodyHandle* h, NbodyLetStats* out) {
    if (!h || !out) return NBODY_ERR_INVALID;
    if (!h->let) return fail(h, NBODY_ERR_INVALID, "not an NBODY_SHARD_SPATIAL handle");
    return nbody::let::stats(h, out);
}

int nbody_debug_let_phase(NbodyHandle* h, int phase, float dt) {
    if (!h) return NBODY_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (!h->let) return fail(h, NBODY_ERR_INVALID, "not an NBODY_SHARD_SPATIAL handle");
    return nbody::let::debug_phase(h, phase, dt);
}

int nbody_debug_let_exchange(NbodyHandle* h, NbodyHandle* peer, int which) {
    if (!h || !peer) return NBODY_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (!h->let || !peer->let) return fail(h, NBODY_ERR_INVALID, "not an NBODY_SHARD_SPATIAL handle");
    return nbody::let::debug_exchange(h, peer, which);
}

int nbody_debug_let_set_prune(NbodyHandle* h, int prune);   // (nbody_let.cpp owns the state)

int nbody_local_range(const NbodyHandle* h, size_t* first, size_t* count) {
    if (!h) return NBODY_ERR_INVALID;
    if (first) *first = h->first_global;
    if (count) *count = h->n_at_upload;
    return NBODY_OK;
}

// ---- test hooks: a sharded step with the exchange done by the caller --------------------------
// Two handles of one process (ranks 0..G-1 of a world of G, all on the same device) can stand in
// for G GPUs: step_begin on each, import every peer's segment into each, step_end on each.  This
// is what nbody_step_by does around the RCCL all-gather; only the transport differs.
int nbody_debug_step_begin(NbodyHandle* h, float dt) {
    if (!h) return NBODY_ERR_INVALID;
    if (h->f64) return fail(h, NBODY_ERR_INVALID, "f64 handles are single-shard");
    int rc = use_device(h);
    return rc ? rc : step_begin(h, dt);
}

int nbody_debug_import_segment(NbodyHandle* h, NbodyHandle* peer) {
    if (!h || !peer) return NBODY_ERR_INVALID;
    if (h->f64 || peer->f64) return fail(h, NBODY_ERR_INVALID, "f64 handles are single-shard");
    if (h->sh.n_seg != peer->sh.n_seg || h->sh.seg_cap != peer->sh.seg_cap) return fail(h, NBODY_ERR_INVALID, "peer has a different sharding");
    int rc = use_device(h);
    if (rc) return rc;
    const int s = peer->sh.my_seg;
    HIP_TRY(h, hipStreamSynchronize(peer->stream));
    HIP_TRY(h, hipMemcpyAsync(h->sh.pos_all + size_t(s) * h->sh.seg_cap, peer->sh.own_pos(), size_t(h->sh.seg_cap) * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->sh.seg_count + s, peer->sh.own_count(), sizeof(int), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->count_dirty = true;
    return NBODY_OK;
}

int nbody_debug_step_forces(NbodyHandle* h, float dt) {
    if (!h) return NBODY_ERR_INVALID;
    if (h->f64) return fail(h, NBODY_ERR_INVALID, "f64 handles are single-shard");
    int rc = use_device(h);
    return rc ? rc : step_forces(h, dt);
}

// what the grouped ncclSend/ncclRecv round delivers: the partial sums `peer` accumulated for this
// shard's bodies, into the plane reserved for that sender
int nbody_debug_import_partials(NbodyHandle* h, NbodyHandle* peer) {
    if (!h || !peer) return NBODY_ERR_INVALID;
    if (h->f64 || peer->f64) return fail(h, NBODY_ERR_INVALID, "f64 handles are single-shard");
    int rc = use_device(h);
    if (rc) return rc;
    if (!(h->cross_on && h->tail_pending)) return NBODY_OK;   // this force pass exchanges nothing
    if (!(peer->cross_on && peer->tail_pending)) return fail(h, NBODY_ERR_INVALID, "peer is not in the same phase");
    int src = -1, dst = -1;
    for (int i = 0; i < peer->cross.parts.n; ++i) if (peer->cross.parts.seg[i] == h->sh.my_seg) src = i;
    for (int i = 0; i < h->cross.n_recv; ++i) if (h->cross.recv_from[i] == peer->sh.my_seg) dst = i;
    if ((src < 0) != (dst < 0)) return fail(h, NBODY_ERR_INVALID, "send/receive plans of the two shards do not match");
    if (src < 0) return NBODY_OK;
    HIP_TRY(h, hipStreamSynchronize(peer->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_planes + size_t(h->recv_plane0 + dst) * h->sym_plan.plane_stride,
                              peer->d_send + size_t(src) * peer->sym_plan.plane_stride,
                              size_t(h->sh.seg_cap) * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return NBODY_OK;
}

int nbody_debug_step_end(NbodyHandle* h, float dt) {
    if (!h) return NBODY_ERR_INVALID;
    if (h->f64) return fail(h, NBODY_ERR_INVALID, "f64 handles are single-shard");
    int rc = use_device(h);
    return rc ? rc : step_finish(h, dt);
}

// Host-only entry (no device needed): which pairs between shards this rank evaluates and whom it
// exchanges partial sums with (kernels_bf_cross.hip make_cross_plan), for tests of the host logic.
// parts: n_parts rows {shard, c0, c1, a0, a1}; recv_from: n_recv ranks in the order their planes are added.
int nbody_host_cross_plan(int rank, int world, int seg_cap, int n_own, int* ipt, int* n_sets, int* n_parts, int* parts,
                          int* n_recv, int* recv_from) {
    if (world < 2 || rank < 0 || rank >= world || seg_cap <= 0 || n_own < 0 || n_own > seg_cap) return NBODY_ERR_INVALID;
    if (world > 2 * (nbody::CrossPartners::kMax - 1)) return NBODY_ERR_INVALID;
    const nbody::CrossPlan p = nbody::make_cross_plan(rank, world, seg_cap, n_own);
    if (ipt) *ipt = p.ipt;
    if (n_sets) *n_sets = p.A;
    if (n_parts) *n_parts = p.parts.n;
    if (parts)
        for (int i = 0; i < p.parts.n; ++i) {
            parts[5 * i + 0] = p.parts.seg[i]; parts[5 * i + 1] = p.parts.c0[i]; parts[5 * i + 2] = p.parts.c1[i];
            parts[5 * i + 3] = p.parts.a0[i]; parts[5 * i + 4] = p.parts.a1[i];
        }
    if (n_recv) *n_recv = p.n_recv;
    if (recv_from) for (int i = 0; i < p.n_recv; ++i) recv_from[i] = p.recv_from[i];
    return NBODY_OK;
}

// Host-only entry (no device needed): where the variable-size rounds of the spatial step (migrants, tree nodes) put their
// messages, from the all-gathered G x G count matrix -- the arithmetic sender and receiver of every pair share.
int nbody_host_launch_plan(size_t n_bodies, float theta2, int fast_math, int out[3]) {
    if (!out) return NBODY_ERR_INVALID;
    nbody::bind_tuning(nullptr);   // (the library's defaults, not some handle's knobs)
    const nbody::WalkPlan p = nbody::walk_plan(n_bodies, fast_math != 0, 64, theta2);
    out[0] = p.bodies_per_lane; out[1] = p.segments; out[2] = nbody::sym_bodies_per_lane(n_bodies);
    return NBODY_OK;
}

int nbody_host_exchange_layout(const int* matrix, int world, int rank, long long clamp, int packed_send, size_t send_stride,
                               size_t* out_at, size_t* n_out, size_t* in_at, size_t* n_in, size_t* total_in) {
    if (!matrix || world < 1 || world > 16 || rank < 0 || rank >= world || !out_at || !n_out || !in_at || !n_in) return NBODY_ERR_INVALID;
    const size_t t = nbody::let::exchange_layout(matrix, world, rank, clamp, packed_send != 0, send_stride, out_at, n_out, in_at, n_in);
    if (total_in) *total_in = t;
    return NBODY_OK;
}

// Host-only entry (no device needed): the octree build alone, for tests of the host logic.
// Arrays hold `cap` nodes (com_mass 4 floats per node); order holds n body ids.
int nbody_host_build_tree(const float* pos4, size_t n, const float center[3], float width, int threads,
                          float* com_mass, float* node_width, int32_t* skip, int32_t* leaf_body, int32_t* order,
                          size_t cap, size_t* n_nodes) {
    if ((!pos4 && n) || !center || !n_nodes) return NBODY_ERR_INVALID;
    if (n > (1ull << 30)) return NBODY_ERR_INVALID;
    // pool, scratch and output arrays persist per calling thread (repeat calls time the build itself)
    static thread_local std::unique_ptr<nbody::WorkerPool> tl_pool;
    static thread_local int tl_threads = 0;
    static thread_local nbody::BuildScratch scratch;
    static thread_local nbody::HostTree tree;
    const int want = threads > 0 ? threads : 1;
    if (!tl_pool || tl_threads != want) { tl_pool.reset(new nbody::WorkerPool(want)); tl_threads = want; }
    nbody::WorkerPool& pool = *tl_pool;
    int cnt = int(n);
    nbody::build_octree(pos4, 1, int(n), &cnt, center, width, pool, scratch, tree);
    if (tree.too_deep) return NBODY_ERR_TREE_DEPTH;
    *n_nodes = tree.n_nodes;
    if (!com_mass) return NBODY_OK;
    if (tree.n_nodes > cap) return NBODY_ERR_CAPACITY;
    for (size_t i = 0; i < tree.n_nodes; ++i) {
        const nbody::NodeRec& r = tree.nodes[i];
        com_mass[4 * i] = r.a.x; com_mass[4 * i + 1] = r.a.y; com_mass[4 * i + 2] = r.a.z; com_mass[4 * i + 3] = r.a.m;
        if (node_width) node_width[i] = std::sqrt(r.b.w2);
        if (skip) skip[i] = r.b.skip;
        if (leaf_body) leaf_body[i] = r.b.body;
    }
    if (order) std::memcpy(order, tree.order, tree.n_order * sizeof(int32_t));
    return NBODY_OK;
}

}  // extern "C"
