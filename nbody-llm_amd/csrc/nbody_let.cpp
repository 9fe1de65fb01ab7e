// nbody_let.cpp -- host orchestration of Barnes-Hut over spatial shards with a halo exchange; see kernels_let.hip for
// the scheme.  A spatial handle is a SINGLE-segment shard (its own bodies only): drift, retain, the device build's
// kernels and the walk run on it as on a one-GPU handle; what is added are five phases separated by four exchanges:
//   phase 0  drift, retain, pick the bodies whose key left the rank's range and pack them per destination
//   --- exchange 0: migrants, variable size, posted with sizes drawn from the last step's counts (the G x G counts travel
//       with them; a step whose migrants do not fit repeats the round exactly: see pass())
//   phase 1  take the immigrants in, keys + sort of the own bodies, first/last key, boxes and weight of the own bodies
//   --- exchange 1: all-gather of the end infos (1.6 KB per rank)
//   phase 2  edge values, delta and the scans of the own slice, contributions to every earlier rank's spanning cells, the
//            keys of the world's quantiles that lie on this rank (next step's bounds)
//   --- exchange 2: all-gather of those tables (13 KB per rank)
//   phase 3  next step's bounds, own slice emitted (local indices), all spanning cells finished, the nodes each partner
//            can reach flagged (ancestors against the boxes its bodies lie in) and written, in node order, into one list
//            per partner
//   --- exchange 3: node records, variable size (counts first: all-gather + the ONE host synchronisation of a step)
//   phase 4  the nodes this rank holds -- its slice and the imports -- laid out in global-index order with their links
//            turned into positions (launch_assemble), walk, kick + half drift
// No buffer here is sized by the world: the slice, the array the walk runs over, the export lists and the staged imports
// follow the rank's own capacity (the last two grow when a step needs more).
// Production (`step`) does the exchanges through the handle's transport on its stream; the tests also run G handles of one
// process on one GPU and do them as device-to-device copies (debug_phase / debug_exchange) -- same kernels, same buffers.
#include "nbody_let.h"
#include "kernels_let.h"

#include <algorithm>
#include <cstring>


namespace nbody {
namespace let {

struct State {
    int G = 1, me = 0;
    bool prune = true;
    bool rebalance = true;        // redraw the ownership bounds every step (NBODY_LET_REBALANCE=0: keep the upload's)
    bool by_work = true;          // ... at the quantiles of the walk's visit counts (NBODY_LET_BALANCE=count: of the body count)
    unsigned long long* d_weight_sum = nullptr;
    unsigned long long* d_bounds_scratch = nullptr;   // [G + 1] where the redrawn bounds go when they are not used
    std::vector<unsigned long long> h_bounds;   // [G + 1]
    unsigned long long* d_bounds = nullptr;
    Migrant* d_send_mig = nullptr;   // [seg_cap] this step's emigrants, contiguous per destination
    Migrant* d_recv_mig = nullptr;   // [mig_recv_cap] this step's immigrants
    size_t mig_recv_cap = 0;
    int mig_in = 0;                  // immigrants of this pass (host: the counts travel first)
    uint64_t migrated_seen = 0;      // bodies_migrated at the last bookkeeping
    unsigned char* d_dest_of = nullptr;   // [seg_cap] destination rank of every own body, 255: stays / left the box
    int *d_send_count = nullptr, *d_send_off = nullptr, *d_mig_cursor = nullptr;   // [G], [G + 1], [G]
    int* d_mig_matrix = nullptr;     // [G][G] row r = rank r's send counts (all-gathered)
    int* d_new_count = nullptr;
    int* d_flags = nullptr;       // [4] sticky flags (also the walk's poison word), step counter, bodies sent away so far
    int* d_box_ord = nullptr;     // [(kBoxes + 1) * 6]
    EndInfo* d_ends = nullptr;    // [G]
    int* d_edge = nullptr;        // [3]
    RoundB* d_rb = nullptr;       // [G]
    int* d_offsets = nullptr;     // [G + 1]
    int* d_top_index = nullptr;   // [G][kLevels]
    int* d_split = nullptr;       // [4]: first[0] = 0, first[1] = total nodes, n_anc[0] = 0
    int local_cap = 0;            // nodes a slice can have
    float4* d_slice = nullptr;    // [local_cap] my slice of the world's node array, links local (the emit writes it)
    float4* d_held = nullptr;     // [held_cap] the nodes this rank holds -- its slice and the imports -- in global-index order,
    size_t held_cap = 0;          //   links = positions: what the walk runs over (launch_assemble)
    float4* d_top_nodes = nullptr;   // [G][kLevels][2] the finished spanning cells (global links)
    unsigned int* d_node_mask = nullptr;   // [local_cap] partners every node of the slice goes to
    int* d_block_n = nullptr;     // [pack_blocks(local_cap)][kMaxRanks]
    int* d_in_n = nullptr;        // [G] records received from each rank this pass
    int* h_in_pin = nullptr;      // pinned staging of d_in_n
    void* d_layout = nullptr;
    int* d_zero = nullptr;        // [1] = 0: the slice is emitted with local indices
    size_t staged = 0;            // emulation: records in the staging buffer so far
    int* d_seg_first = nullptr;   // [kWalkMaxSplit + 1] first node of every walk segment | [kWalkMaxSplit] ancestor counts | [kWalkMaxSplit][192] ancestors
    float4* d_walk_planes = nullptr;   // [segments][stride] the segments' partial accelerations
    size_t walk_planes_cap = 0;
    uint64_t send_regrown = 0;    // times the export buffer had to grow (and the lists were written again)
    int* d_order = nullptr;
    int* d_tree_info = nullptr;   // [4] of the local build
    void* d_ws = nullptr;         // workspace of the build
    size_t ws_cap = 0;
    int* d_parent = nullptr;
    unsigned char* d_depth = nullptr;
    unsigned int* d_upper_ok = nullptr;   // [kLevels]
    int2* d_node_flags = nullptr;   // per node of the slice: {its parent, the partners that could open it}
    int* d_let_count = nullptr;   // [G] nodes for each partner
    LetRecord* d_let_send = nullptr;      // [let_send_cap] the export lists, one after the other (d_list_first)
    size_t let_send_cap = 0;
    int* d_list_first = nullptr;          // [G + 1]
    LetRecord* d_let_recv = nullptr;
    size_t let_recv_cap = 0;
    int* d_let_matrix = nullptr;  // [G][G] counts (all-gathered rows)
    int* h_pin = nullptr;         // pinned scratch [report_ints(G) + 64]
    int* d_report = nullptr;      // [report_ints(G)] what the host reads once per step (k_let_report)
    // the migrant round with sizes posted ahead of the counts: pred = what every message is posted with, drawn from the
    // last step's counts (the same on every rank); a step whose migrants do not fit makes the round again, exactly
    bool have_pred = false;
    std::vector<int> h_pred;      // [G * G]
    int* d_pred = nullptr;        // [G * G]
    int* h_pred_pin = nullptr;    // pinned staging of d_pred
    Migrant* d_slot_out = nullptr;   // the packed emigrants again, one slot of pred[me][r] records per destination
    size_t slot_out_cap = 0;
    long long count_delta = 0;    // bodies pushed / removed since the last pass (count_global reads the last pass's table)
    uint64_t spills = 0;          // steps that had to make their migrant round twice
    uint64_t host_syncs = 0;      // host synchronisations inside passes (NbodyLetStats.host_syncs)
    uint64_t node_array_peak = 0; // most node records this rank held at once: own slice + imports
    TreeDevWork work;
    NbodyLetStats st{};
    std::vector<int> recv_n;      // emulation / production: records received from each rank this pass
    hipEvent_t ev[2][10] = {};    // begin/end of the five phases (nbody_set_profiling), two sets: a step's are read during the next
    bool ev_made = false, ev_live[2] = {false, false};
    int ev_set = 0;               // the set the pass under way records into
};

namespace {

constexpr long long kNoClamp = 0x7fffffffLL;   // the lists travel whole (exchange_layout's clamp)
constexpr int kWalkMaxSplit = 64;              // node-range segments of the walk (kernels_bh.hip WalkSplit), at most

int fail(NbodyHandle* h, int code, const std::string& msg) { h->err = msg; return code; }

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

// phase timing: an event at both ends of a phase, on the handle's stream; account() adds the five durations up
struct PhaseTimer {
    NbodyHandle* h; State& s; int phase;
    PhaseTimer(NbodyHandle* h_, State& s_, int p) : h(h_), s(s_), phase(p) {
        if (!h->profiling) { if (p == 0) s.ev_live[s.ev_set] = false; return; }
        if (!s.ev_made) { for (auto& set : s.ev) for (hipEvent_t& e : set) (void)hipEventCreate(&e); s.ev_made = true; }
        if (p == 0) s.ev_live[s.ev_set] = true;
        if (s.ev_live[s.ev_set]) (void)hipEventRecord(s.ev[s.ev_set][2 * p], h->stream);
    }
    ~PhaseTimer() { if (h->profiling && s.ev_live[s.ev_set]) (void)hipEventRecord(s.ev[s.ev_set][2 * phase + 1], h->stream); }
};

// adds up the five durations of a finished pass (the caller knows its events have completed)
void collect_phase_times(State& s, int set) {
    if (!s.ev_live[set]) return;
    for (int p = 0; p < 5; ++p) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.ev[set][2 * p], s.ev[set][2 * p + 1]) == hipSuccess) s.st.phase_ms[p] += double(ms);
    }
    s.ev_live[set] = false;
}

unsigned long long host_key(const float* p, const float c[3], float width) {   // kernels_tree.hip k_tree_keys on the host
    float cx = c[0], cy = c[1], cz = c[2];
    float hw = width * 0.5f;
    unsigned long long key = 0;
    for (int l = 0; l < kLevels; ++l) {
        const bool bx = p[0] > cx, by = p[1] > cy, bz = p[2] > cz;
        key = (key << 3) | (unsigned long long)((bx ? 1 : 0) | (by ? 2 : 0) | (bz ? 4 : 0));
        hw = hw * 0.5f;
        cx = bx ? cx + hw : cx - hw;
        cy = by ? cy + hw : cy - hw;
        cz = bz ? cz + hw : cz - hw;
    }
    return key;
}

template <class T> int dev_alloc(NbodyHandle* h, T** p, size_t n) {
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(1, n) * sizeof(T)));
    HIP_TRY(h, hipMemsetAsync(*p, 0, std::max<size_t>(1, n) * sizeof(T), h->stream));
    return NBODY_OK;
}

int ensure_node_buffers(NbodyHandle* h, State& s) {
    const Shard& sh = h->sh;
    const int want_local = 4 * sh.seg_cap + 64;
    if (s.local_cap < want_local) {
        for (void* p : {(void*)s.d_parent, (void*)s.d_depth, (void*)s.d_node_flags, (void*)s.d_let_send, (void*)s.d_slice, (void*)s.d_node_mask, (void*)s.d_block_n})
            if (p) (void)hipFree(p);
        s.d_parent = nullptr; s.d_depth = nullptr; s.d_node_flags = nullptr; s.d_let_send = nullptr; s.d_slice = nullptr; s.d_node_mask = nullptr; s.d_block_n = nullptr;
        int rc;
        if ((rc = dev_alloc(h, &s.d_slice, size_t(want_local) * 2))) return rc;
        if ((rc = dev_alloc(h, &s.d_node_mask, size_t(want_local)))) return rc;
        if ((rc = dev_alloc(h, &s.d_block_n, pack_blocks(want_local) * size_t(kMaxRanks)))) return rc;
        if ((rc = dev_alloc(h, &s.d_parent, size_t(want_local)))) return rc;
        if ((rc = dev_alloc(h, &s.d_depth, size_t(want_local)))) return rc;
        if ((rc = dev_alloc(h, &s.d_node_flags, size_t(want_local)))) return rc;
        // exports and imports: room for a quarter of the slice's node capacity each -- the body capacity's worth of nodes, several
        // times what the tests and the configs' sizes move (0.2-0.7 of the slice's LIVE nodes); both grow when a step needs more
        const size_t lists = size_t(want_local) / size_t(std::max(1, tuning().let_list_div)) + 1024;
        s.let_send_cap = lists;
        if ((rc = dev_alloc(h, &s.d_let_send, s.let_send_cap))) return rc;
        if (s.let_recv_cap < lists) {
            if (s.d_let_recv) (void)hipFree(s.d_let_recv);
            s.d_let_recv = nullptr; s.let_recv_cap = 0;
            if ((rc = dev_alloc(h, &s.d_let_recv, lists))) return rc;
            s.let_recv_cap = lists;
        }
        s.local_cap = want_local;
    }
    // what the walk runs over: room for the slice and for as many imports as the staging buffer takes -- a rank's own
    // capacity decides it, not the world's size
    if (s.held_cap < size_t(s.local_cap) + s.let_recv_cap) {
        if (s.d_held) (void)hipFree(s.d_held);
        s.d_held = nullptr; s.held_cap = 0;
        int rc;
        if ((rc = dev_alloc(h, &s.d_held, (size_t(s.local_cap) + s.let_recv_cap) * 2))) return rc;
        s.held_cap = size_t(s.local_cap) + s.let_recv_cap;
    }
    if (s.ws_cap < size_t(sh.seg_cap)) {
        if (s.d_ws) (void)hipFree(s.d_ws);
        if (s.d_order) (void)hipFree(s.d_order);
        if (s.d_send_mig) (void)hipFree(s.d_send_mig);
        if (s.d_dest_of) (void)hipFree(s.d_dest_of);
        s.d_ws = nullptr; s.d_order = nullptr; s.d_send_mig = nullptr; s.d_dest_of = nullptr;
        {
            int rc;
            if ((rc = dev_alloc(h, &s.d_send_mig, size_t(sh.seg_cap)))) return rc;
            if ((rc = dev_alloc(h, &s.d_dest_of, size_t(sh.seg_cap)))) return rc;
        }
        HIP_TRY(h, hipMalloc(&s.d_ws, tree_build_workspace_bytes(size_t(sh.seg_cap))));
        int rc;
        if ((rc = dev_alloc(h, &s.d_order, size_t(sh.seg_cap)))) return rc;
        s.ws_cap = size_t(sh.seg_cap);
    }
    return NBODY_OK;
}

// room for `need` immigrants; the first `keep` records of the old buffer survive a reallocation
int ensure_mig_recv(NbodyHandle* h, State& s, size_t need, size_t keep) {
    if (need <= s.mig_recv_cap) return NBODY_OK;
    const size_t cap = need + need / 2 + 1024;
    Migrant* fresh = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&fresh), cap * sizeof(Migrant)));
    if (keep > 0 && s.d_recv_mig) HIP_TRY(h, hipMemcpyAsync(fresh, s.d_recv_mig, keep * sizeof(Migrant), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (s.d_recv_mig) (void)hipFree(s.d_recv_mig);
    s.d_recv_mig = fresh;
    s.mig_recv_cap = cap;
    return NBODY_OK;
}

// room for `need` imported node records in the staging buffer (the first `keep` survive a reallocation), and with it in
// the array the walk runs over.  Called at points where the host is synchronised with the stream or about to be.
int ensure_let_recv(NbodyHandle* h, State& s, size_t need, size_t keep) {
    if (need <= s.let_recv_cap) return NBODY_OK;
    const size_t cap = need + need / 4 + 1024;
    LetRecord* fresh = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&fresh), cap * sizeof(LetRecord)));
    if (keep > 0 && s.d_let_recv) HIP_TRY(h, hipMemcpyAsync(fresh, s.d_let_recv, keep * sizeof(LetRecord), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (s.d_let_recv) (void)hipFree(s.d_let_recv);
    s.d_let_recv = fresh;
    s.let_recv_cap = cap;
    if (s.d_held) (void)hipFree(s.d_held);
    s.d_held = nullptr; s.held_cap = 0;
    int rc = dev_alloc(h, &s.d_held, (size_t(s.local_cap) + s.let_recv_cap) * 2);
    if (rc) return rc;
    s.held_cap = size_t(s.local_cap) + s.let_recv_cap;
    return NBODY_OK;
}

// room for `need` records of export lists; growing means writing the lists again (launch_pack: the counts and every node's
// partners are still on the device).  Called where the host has just read this pass's counts.
int ensure_let_send(NbodyHandle* h, State& s, size_t need) {
    if (need <= s.let_send_cap) return NBODY_OK;
    if (s.d_let_send) (void)hipFree(s.d_let_send);
    s.d_let_send = nullptr; s.let_send_cap = 0;
    const size_t cap = need + need / 4 + 1024;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&s.d_let_send), cap * sizeof(LetRecord)));
    s.let_send_cap = cap;
    s.send_regrown += 1;
    launch_pack(h->stream, s.local_cap, s.d_tree_info, s.d_slice, s.d_top_nodes, s.d_offsets, s.d_top_index, s.d_depth, s.d_node_mask, s.d_block_n,
                s.d_list_first, s.G, s.me, s.d_let_send, s.let_send_cap);
    HIP_TRY(h, hipGetLastError());
    return NBODY_OK;
}

// ---- the phases (everything is enqueued on h->stream; nothing here waits for the device)
int phase0(NbodyHandle* h, State& s, float dt, bool drift) {
    Shard& sh = h->sh;
    PhaseTimer timer(h, s, 0);
    int rc = ensure_node_buffers(h, s);
    if (rc) return rc;
    if (drift) {
        if (!h->bounds_set) return fail(h, NBODY_ERR_INVALID, "nbody_set_bounds has not been called");
        launch_drift_half(h->stream, sh, int(h->n_local), dt, h->bnd);   // integrate_pre_force
    }
    launch_classify(h->stream, sh, int(h->n_local), h->center, h->width, s.d_bounds, s.G, s.me, s.d_dest_of, s.d_send_mig, s.d_send_count,
                    s.d_send_off, s.d_mig_cursor, drift);
    s.mig_in = 0;
    launch_compact(h->stream, sh, int(h->n_local));                      // retain, and the emigrants leave
    h->count_dirty = true;
    HIP_TRY(h, hipGetLastError());
    return NBODY_OK;
}

// slots_in >= 0: the immigrants sit in receive slots of the predicted sizes (launch_append_slots); -1: packed, s.mig_in of them
int phase1(NbodyHandle* h, State& s, int slots_in = -1) {
    Shard& sh = h->sh;
    PhaseTimer timer(h, s, 1);
    if (slots_in >= 0) launch_append_slots(h->stream, sh, s.d_recv_mig, slots_in, s.d_mig_matrix, s.d_pred, s.G, s.me, s.d_flags, s.d_new_count, s.d_send_count);
    else launch_append(h->stream, sh, s.d_recv_mig, s.mig_in, s.G, s.d_flags, s.d_new_count, s.d_send_count);
    // the host's bound of the own count: what was there before the retain + what arrived
    h->n_local = std::min<size_t>(size_t(sh.seg_cap), h->n_local + size_t(s.mig_in));
    if (tree_sort_keys(h->stream, sh.own_pos(), sh.own_count(), int(h->n_local), h->center, h->width, s.d_ws, s.ws_cap, s.d_tree_info, &s.work) != 0)
        return fail(h, NBODY_ERR_HIP, "device octree build: rocPRIM call failed");
    launch_ends(h->stream, sh, int(h->n_local), s.work.keys, s.work.ids, s.d_box_ord, s.d_weight_sum, s.d_ends + s.me);
    HIP_TRY(h, hipGetLastError());
    return NBODY_OK;
}

int phase2(NbodyHandle* h, State& s) {
    Shard& sh = h->sh;
    PhaseTimer timer(h, s, 2);
    launch_edges(h->stream, s.d_ends, s.G, s.me, s.d_edge);
    // delta, the scans: how many nodes my slice has, every body's first node, the prefix sums the spanning cells need
    if (tree_scan_sorted(h->stream, sh.own_pos(), sh.own_count(), int(h->n_local), s.d_ws, s.ws_cap, s.d_tree_info, s.d_edge, sh.acc) != 0)
        return fail(h, NBODY_ERR_HIP, "device octree build failed");
    launch_contrib(h->stream, sh, s.work, s.d_tree_info, s.d_ends, s.d_edge, s.G, s.me, s.d_rb + s.me, s.by_work, s.d_flags);
    HIP_TRY(h, hipGetLastError());
    return NBODY_OK;
}

int phase3(NbodyHandle* h, State& s) {
    Shard& sh = h->sh;
    PhaseTimer timer(h, s, 3);
    HIP_TRY(h, hipMemsetAsync(s.d_let_count, 0, sizeof(int) * s.G, h->stream));
    launch_offsets(h->stream, s.d_rb, s.G, s.d_offsets, s.d_flags, s.rebalance ? s.d_bounds : s.d_bounds_scratch);
    // my slice of the world's node array, with local indices and links (the wire and the assembly shift them)
    if (tree_emit_nodes(h->stream, sh.own_pos(), sh.own_count(), int(h->n_local), h->width, s.d_ws, s.ws_cap, s.d_slice, s.local_cap, s.local_cap,
                        s.d_order, s.d_tree_info, 0, s.d_edge, s.d_zero, s.d_parent, s.d_depth) != 0)
        return fail(h, NBODY_ERR_HIP, "device octree build failed");
    launch_finalize(h->stream, s.d_rb, s.d_ends, s.G, s.me, h->width, s.d_slice, s.d_offsets, s.d_top_index, s.d_top_nodes);
    launch_flags_and_pack(h->stream, s.local_cap, s.d_tree_info, s.d_edge, s.d_slice, s.d_top_nodes, s.d_offsets, s.d_top_index, s.d_ends, s.G, s.me,
                          h->theta2, s.d_parent, s.d_depth, s.d_upper_ok, s.d_node_flags, s.d_node_mask, s.d_block_n, s.d_let_count, s.d_list_first,
                          s.d_let_send, s.let_send_cap, s.prune);
    HIP_TRY(h, hipGetLastError());
    return NBODY_OK;
}

// the walk over the assembled array (+ kick + half drift when this is a step)
int phase4(NbodyHandle* h, State& s, float dt, bool kick) {
    Shard& sh = h->sh;
    PhaseTimer timer(h, s, 4);
    // the walk's shape for this many bodies (kernels.h walk_plan): bodies per lane inside launch_bh_walk, node-range segments here
    // -- a rank with 10^5-10^6 bodies is a few thousand waves at one segment, not enough to fill the chip
    const bool fast = h->cfg.math_mode != NBODY_MATH_STRICT;
    int K = h->n_local > 0 ? walk_plan(h->n_local, fast, kWalkMaxSplit, h->theta2).segments : 1;
    while (K > 1 && size_t(K) * 16 > h->n_local) K /= 2;   // (a tree has at least as many nodes as bodies; tiny ranks walk in one piece)
    // the nodes this rank holds, in global-index order, links = positions; first[] of the unsplit walk = {0, nodes held}
    {
        size_t staged_upper = 0;
        for (int r = 0; r < s.G; ++r) { s.h_in_pin[r] = s.recv_n[r]; staged_upper += size_t(std::max(0, s.recv_n[r])); }
        HIP_TRY(h, hipMemcpyAsync(s.d_in_n, s.h_in_pin, sizeof(int) * size_t(s.G), hipMemcpyHostToDevice, h->stream));
        launch_assemble(h->stream, s.d_slice, s.local_cap, s.d_let_recv, int(staged_upper), s.d_in_n, s.d_tree_info, s.d_offsets, s.d_top_index,
                        s.d_top_nodes, s.G, s.me, s.d_layout, s.d_split, s.d_held, K, s.d_seg_first);
    }
    TreeDev td;
    td.nodes = s.d_held; td.n_nodes = int(std::min<size_t>(s.held_cap, 0x7fffffff));
    td.order = s.d_order; td.n_order = int(h->n_local);
    td.n_order_dev = sh.own_count();
    td.poison = s.d_flags;
    td.store_work = 1;
    td.n_split = K;
    if (K > 1) {
        const size_t stride = (h->n_local + 1023) / 1024 * 1024;
        if (size_t(K) * stride > s.walk_planes_cap) {
            if (s.d_walk_planes) (void)hipFree(s.d_walk_planes);
            s.d_walk_planes = nullptr; s.walk_planes_cap = 0;
            const size_t want = size_t(K) * stride + size_t(K) * stride / 8;
            HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&s.d_walk_planes), want * sizeof(float4)));
            s.walk_planes_cap = want;
        }
        // the segments start at global node indices total k / K (launch_assemble found where those fall in the held array); their
        // ancestors come from the held array itself (the local build's arrays know the slice only)
        launch_walk_split_scan(h->stream, s.d_held, s.d_split, K, s.d_seg_first, s.d_seg_first + kWalkMaxSplit + 1, s.d_seg_first + 2 * kWalkMaxSplit + 1, 1);
        td.split_first = s.d_seg_first;
        td.split_n_anc = s.d_seg_first + kWalkMaxSplit + 1;
        td.split_anc = s.d_seg_first + 2 * kWalkMaxSplit + 1;
        td.split_planes = s.d_walk_planes;
        td.split_stride = stride;
    } else {
        td.split_first = s.d_split;          // {0, nodes held} (k_let_layout)
        td.split_n_anc = s.d_split + 2;      // 0
        td.split_anc = s.d_split + 2;
    }
    int kicked = 0;
    launch_bh_walk(h->stream, sh, td, h->g, h->g_soft * h->g_soft, h->theta2, fast ? 1 : 0, h->d_counters, h->cfg.leaf_mode == NBODY_LEAF_DIRECT,
                   kick ? &dt : nullptr, &kicked);
    if (kick) {
        if (!kicked) launch_kick_drift(h->stream, sh, int(h->n_local), dt);   // integrate_after_force (K > 1: it rode in the plane reduction)
        h->elapsed += dt;
        h->stats.steps += 1;
    }
    HIP_TRY(h, hipGetLastError());
    s.st.steps += 1;
    return NBODY_OK;
}

}  // namespace

int create(NbodyHandle* h) {
    State* sp = new State();
    h->let = sp;
    State& s = *sp;
    s.G = h->cfg.world_size;
    s.me = h->cfg.rank;
    if (s.G > kMaxRanks) return fail(h, NBODY_ERR_INVALID, "NBODY_SHARD_SPATIAL supports up to 16 ranks");
    s.recv_n.assign(s.G, 0);
    int rc;
    if ((rc = dev_alloc(h, &s.d_bounds, size_t(s.G) + 1))) return rc;
    if ((rc = dev_alloc(h, &s.d_bounds_scratch, size_t(s.G) + 1))) return rc;
    if (const char* v = std::getenv("NBODY_LET_REBALANCE")) s.rebalance = std::atoi(v) != 0;
    if (const char* v = std::getenv("NBODY_LET_BALANCE")) s.by_work = std::strcmp(v, "count") != 0;
    if ((rc = dev_alloc(h, &s.d_weight_sum, 1))) return rc;
    if ((rc = dev_alloc(h, &s.d_send_count, size_t(s.G)))) return rc;
    if ((rc = dev_alloc(h, &s.d_send_off, size_t(s.G) + 1))) return rc;
    if ((rc = dev_alloc(h, &s.d_mig_cursor, size_t(s.G)))) return rc;
    if ((rc = dev_alloc(h, &s.d_mig_matrix, size_t(s.G) * s.G))) return rc;
    if ((rc = dev_alloc(h, &s.d_new_count, 1))) return rc;
    if ((rc = dev_alloc(h, &s.d_flags, 4))) return rc;
    if ((rc = dev_alloc(h, &s.d_box_ord, (kBoxes + 1) * 6))) return rc;
    {
        std::vector<int> init((kBoxes + 1) * 6);
        for (size_t t = 0; t < init.size(); ++t) init[t] = (t % 6) < 3 ? 0x7fffffff : int(0x80000000);
        HIP_TRY(h, hipMemcpyAsync(s.d_box_ord, init.data(), init.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    if ((rc = dev_alloc(h, &s.d_ends, size_t(s.G)))) return rc;
    if ((rc = dev_alloc(h, &s.d_edge, 4))) return rc;
    if ((rc = dev_alloc(h, &s.d_rb, size_t(s.G)))) return rc;
    if ((rc = dev_alloc(h, &s.d_offsets, size_t(s.G) + 1))) return rc;
    if ((rc = dev_alloc(h, &s.d_top_index, size_t(s.G) * kLevels))) return rc;
    if ((rc = dev_alloc(h, &s.d_split, 4))) return rc;
    if ((rc = dev_alloc(h, &s.d_tree_info, 4))) return rc;
    if ((rc = dev_alloc(h, &s.d_upper_ok, kLevels))) return rc;
    if ((rc = dev_alloc(h, &s.d_let_count, size_t(s.G)))) return rc;
    if ((rc = dev_alloc(h, &s.d_let_matrix, size_t(s.G) * s.G))) return rc;
    if ((rc = dev_alloc(h, &h->sh.ids, size_t(h->sh.seg_cap)))) return rc;
    HIP_TRY(h, hipHostMalloc(&s.h_pin, (size_t(report_ints(s.G)) + 64) * sizeof(int), hipHostMallocDefault));
    if ((rc = dev_alloc(h, &s.d_report, size_t(report_ints(s.G))))) return rc;
    if ((rc = dev_alloc(h, &s.d_pred, size_t(s.G) * s.G))) return rc;
    if ((rc = dev_alloc(h, &s.d_top_nodes, size_t(s.G) * kLevels * 2))) return rc;
    if ((rc = dev_alloc(h, &s.d_list_first, size_t(s.G) + 1))) return rc;
    if ((rc = dev_alloc(h, &s.d_seg_first, size_t(2 * kWalkMaxSplit + 1 + kWalkMaxSplit * 192)))) return rc;
    if ((rc = dev_alloc(h, &s.d_in_n, size_t(s.G)))) return rc;
    if ((rc = dev_alloc(h, &s.d_zero, 1))) return rc;
    if ((rc = dev_alloc(h, reinterpret_cast<char**>(&s.d_layout), layout_bytes()))) return rc;
    HIP_TRY(h, hipHostMalloc(&s.h_in_pin, size_t(s.G) * sizeof(int), hipHostMallocDefault));
    HIP_TRY(h, hipHostMalloc(&s.h_pred_pin, size_t(s.G) * s.G * sizeof(int), hipHostMallocDefault));
    s.h_pred.assign(size_t(s.G) * s.G, 0);
    h->sh.poison = s.d_flags;   // a raised flag stops every kernel that would change the state
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return NBODY_OK;
}

void destroy(NbodyHandle* h) {
    State* s = h->let;
    if (!s) return;
    void* dev[] = {s->d_bounds, s->d_bounds_scratch, s->d_weight_sum, s->d_send_mig, s->d_recv_mig, s->d_send_count, s->d_send_off, s->d_mig_cursor, s->d_mig_matrix, s->d_dest_of, s->d_new_count, s->d_flags, s->d_box_ord,
                   s->d_ends, s->d_edge, s->d_rb, s->d_offsets, s->d_top_index, s->d_split, s->d_slice, s->d_held, s->d_top_nodes, s->d_node_mask,
                   s->d_block_n, s->d_in_n, s->d_layout, s->d_zero, s->d_list_first, s->d_seg_first, s->d_walk_planes, s->d_order, s->d_tree_info,
                   s->d_ws, s->d_parent, s->d_depth, s->d_upper_ok, s->d_node_flags, s->d_let_count, s->d_let_send,
                   s->d_let_recv, s->d_let_matrix, h->sh.ids, s->d_report, s->d_pred, s->d_slot_out};
    for (void* p : dev) if (p) (void)hipFree(p);
    h->sh.ids = nullptr;
    h->sh.poison = nullptr;
    if (s->h_pin) (void)hipHostFree(s->h_pin);
    if (s->h_pred_pin) (void)hipHostFree(s->h_pred_pin);
    if (s->h_in_pin) (void)hipHostFree(s->h_in_pin);
    if (s->ev_made) for (auto& set : s->ev) for (hipEvent_t e : set) (void)hipEventDestroy(e);
    delete s;
    h->let = nullptr;
}

// Every rank passes the same full vector.  Bounds = the G-quantiles of the keys; this rank keeps the bodies of its range.
int upload(NbodyHandle* h, const void* aos, size_t n, size_t stride) {
    State& s = *h->let;
    Shard& sh = h->sh;
    if (!h->bounds_set) return fail(h, NBODY_ERR_INVALID, "NBODY_SHARD_SPATIAL: nbody_set_bounds must come before nbody_upload (the shards are key ranges of the root box)");
    const char* src = static_cast<const char*>(aos);
    std::vector<unsigned long long> key(n);
    std::vector<uint32_t> idx(n);
    for (size_t k = 0; k < n; ++k) {
        float p[3];
        std::memcpy(p, src + k * stride, 12);
        key[k] = host_key(p, h->center, h->width);
        idx[k] = uint32_t(k);
    }
    std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return key[a] != key[b] ? key[a] < key[b] : a < b; });
    s.h_bounds.assign(size_t(s.G) + 1, 0ull);
    for (int r = 1; r < s.G; ++r) s.h_bounds[r] = n ? key[idx[std::min(n - 1, size_t(r) * n / size_t(s.G))]] : 0ull;
    for (int r = 1; r < s.G; ++r) s.h_bounds[r] = std::max(s.h_bounds[r], s.h_bounds[r - 1]);
    s.h_bounds[s.G] = 1ull << 63;
    HIP_TRY(h, hipMemcpyAsync(s.d_bounds, s.h_bounds.data(), (size_t(s.G) + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
    // own bodies, in the vector's order
    std::vector<int> own;
    for (size_t k = 0; k < n; ++k) {
        int dest = 0;
        for (int r = 1; r < s.G; ++r) if (s.h_bounds[r] <= key[k]) dest = r;
        if (dest == s.me) own.push_back(int(k));
    }
    if (own.size() > size_t(sh.seg_cap)) return fail(h, NBODY_ERR_CAPACITY, "NBODY_SHARD_SPATIAL: this rank's key range holds more bodies than its capacity");
    const size_t m = own.size();
    // staging through the handle's AoS buffers (40-byte records)
    if (m > h->aos_cap) {
        if (h->d_aos) (void)hipFree(h->d_aos);
        if (h->h_aos) (void)hipHostFree(h->h_aos);
        h->d_aos = nullptr; h->h_aos = nullptr; h->aos_cap = 0;
        HIP_TRY(h, hipMalloc(&h->d_aos, std::max<size_t>(1, m) * 40));
        HIP_TRY(h, hipHostMalloc(&h->h_aos, std::max<size_t>(1, m) * 40, hipHostMallocDefault));
        h->aos_cap = std::max<size_t>(1, m);
    }
    for (size_t j = 0; j < m; ++j) std::memcpy(h->h_aos + 10 * j, src + size_t(own[j]) * stride, 40);
    if (m) HIP_TRY(h, hipMemcpyAsync(h->d_aos, h->h_aos, m * 40, hipMemcpyHostToDevice, h->stream));
    launch_aos_to_soa(h->stream, h->d_aos, 10, int(m), sh.own_pos(), sh.vel, sh.acc);
    if (m) HIP_TRY(h, hipMemcpyAsync(sh.ids, own.data(), m * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(sh.escaped, 0, sizeof(int), h->stream));
    HIP_TRY(h, hipMemsetAsync(s.d_send_count, 0, sizeof(int) * s.G, h->stream));
    HIP_TRY(h, hipMemsetAsync(s.d_flags, 0, 4 * sizeof(int), h->stream));
    s.have_pred = false;   // (the first step after an upload learns the migrant counts the slow way)
    h->n_local = m;
    h->seg_count_host[0] = int(m);
    h->h_counts[0] = int(m);
    HIP_TRY(h, hipMemcpyAsync(sh.seg_count, h->h_counts, sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));   // own[] and the staging are reused
    h->count_dirty = false;
    h->first_global = 0; h->n_at_upload = m;
    return NBODY_OK;
}

int download_ids(NbodyHandle* h, int32_t* ids, size_t cap, size_t* n_out) {
    Shard& sh = h->sh;
    HIP_TRY(h, hipMemcpyAsync(h->h_counts, sh.seg_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t n = size_t(h->h_counts[0]);
    h->n_local = n; h->seg_count_host[0] = int(n); h->count_dirty = false;
    if (n_out) *n_out = n;
    if (n > cap) return fail(h, NBODY_ERR_CAPACITY, "download buffer too small");
    if (n && ids) {
        HIP_TRY(h, hipMemcpy(ids, sh.ids, n * sizeof(int), hipMemcpyDeviceToHost));
    }
    return NBODY_OK;
}

int check_flags(NbodyHandle* h) {
    State& s = *h->let;
    HIP_TRY(h, hipMemcpyAsync(s.h_pin, s.d_flags, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int f = s.h_pin[0];
    if (!f) return NBODY_OK;
    std::string who;   // which ranks' builds raised something (their RoundB records are here)
    {
        std::vector<RoundB> rb(size_t(s.G));
        if (hipMemcpy(rb.data(), s.d_rb, sizeof(RoundB) * s.G, hipMemcpyDeviceToHost) == hipSuccess)
            for (int q = 0; q < s.G; ++q) if (rb[q].flags) who += " rank " + std::to_string(q) + ": build flags " + std::to_string(rb[q].flags) + ";";
        who += " flag word " + std::to_string(f);
    }
    // (an unsorted clump also looks "too deep" to the passes behind the sort: the clump is the cause)
    if (f & kFlagBigGroup) return fail(h, NBODY_ERR_TREE_DEPTH, "spatial shards: more than 4096 bodies of one rank share 16 levels of the tree: beyond what the device build sorts;" + who);
    if (f & (kFlagDeep)) return fail(h, NBODY_ERR_TREE_DEPTH, "spatial shards: two bodies separate only below the device build's 42 levels (coincident?);" + who);
    if (f & kFlagBigGroup) return fail(h, NBODY_ERR_TREE_DEPTH, "spatial shards: more than 4096 bodies of one rank share 16 levels of the tree (a clump inside a cell 1.5e-5 of the box wide): beyond what the device build sorts, and a spatial rank has no host build to hand the step to" + who);
    if (f & kFlagCapacity) return fail(h, NBODY_ERR_CAPACITY, "spatial shards: a rank's capacity is exhausted by immigrants");
    if (f & (kFlagNodeCap | kFlagNodeCapLocal)) return fail(h, NBODY_ERR_CAPACITY, "spatial shards: node array too small");
    return fail(h, NBODY_ERR_INVALID, "spatial shards: the device raised flag " + std::to_string(f));
}

int count_global(NbodyHandle* h, size_t* n_out) {
    State& s = *h->let;
    std::vector<EndInfo> e(size_t(s.G));
    HIP_TRY(h, hipMemcpy(e.data(), s.d_ends, sizeof(EndInfo) * s.G, hipMemcpyDeviceToHost));
    size_t t = 0;
    for (const EndInfo& x : e) t += size_t(std::max(0, x.n_bodies));
    *n_out = size_t(std::max<long long>(0, (long long)t + s.count_delta));
    return NBODY_OK;
}

// ---- Clone / Vec::push / Vec::swap_remove on spatial shards (include/nbody_hip.h) -----------------------------------
int clone_state(NbodyHandle* src, NbodyHandle* dst) {
    State& a = *src->let;
    State& b = *dst->let;
    HIP_TRY(src, hipStreamSynchronize(src->stream));
    const size_t cap = size_t(src->sh.seg_cap);
    HIP_TRY(dst, hipMemcpyAsync(dst->sh.ids, src->sh.ids, cap * sizeof(int), hipMemcpyDeviceToDevice, dst->stream));
    HIP_TRY(dst, hipMemcpyAsync(b.d_bounds, a.d_bounds, (size_t(a.G) + 1) * sizeof(unsigned long long), hipMemcpyDeviceToDevice, dst->stream));
    HIP_TRY(dst, hipMemcpyAsync(b.d_ends, a.d_ends, size_t(a.G) * sizeof(EndInfo), hipMemcpyDeviceToDevice, dst->stream));   // (count_global reads it)
    HIP_TRY(dst, hipStreamSynchronize(dst->stream));
    b.h_bounds = a.h_bounds;
    b.prune = a.prune; b.rebalance = a.rebalance; b.by_work = a.by_work;
    b.count_delta = a.count_delta;
    b.have_pred = false;
    return NBODY_OK;
}

namespace {
// exact own count + the own ids on the host (ascending copy in `sorted`)
int fetch_ids(NbodyHandle* h, std::vector<int>& ids, std::vector<int>& sorted) {
    Shard& sh = h->sh;
    HIP_TRY(h, hipMemcpyAsync(h->h_counts, sh.seg_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->n_local = size_t(h->h_counts[0]); h->seg_count_host[0] = int(h->n_local); h->count_dirty = false;
    ids.resize(h->n_local);
    if (h->n_local) HIP_TRY(h, hipMemcpy(ids.data(), sh.ids, h->n_local * sizeof(int), hipMemcpyDeviceToHost));
    sorted = ids;
    std::sort(sorted.begin(), sorted.end());
    return NBODY_OK;
}
// one number from every rank (host side; a world of one without a communicator is just itself)
int gather_ll(NbodyHandle* h, long long mine, std::vector<long long>& all) {
    State& s = *h->let;
    all.assign(size_t(s.G), 0);
    if (!h->comm_ready) {
        if (s.G > 1) return fail(h, NBODY_ERR_COMM, "world_size > 1 but nbody_comm_init has not been called");
        all[0] = mine;
        return NBODY_OK;
    }
    TP_TRY(h, h->tp->host_all_gather(&mine, all.data(), sizeof(long long)));
    return NBODY_OK;
}
int push_count(NbodyHandle* h) {
    h->h_counts[0] = int(h->n_local);
    h->seg_count_host[0] = int(h->n_local);
    HIP_TRY(h, hipMemcpyAsync(h->sh.seg_count, h->h_counts, sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return NBODY_OK;
}
}  // namespace

// Vec::push: the body goes to the rank whose key range holds it; its place in the vector is the next free index
int add_point(NbodyHandle* h, const void* particle) {
    State& s = *h->let;
    Shard& sh = h->sh;
    if (!h->bounds_set) return fail(h, NBODY_ERR_INVALID, "nbody_set_bounds has not been called");
    std::vector<int> ids, sorted;
    int rc = fetch_ids(h, ids, sorted);
    if (rc) return rc;
    std::vector<long long> tops;
    rc = gather_ll(h, sorted.empty() ? -1LL : (long long)sorted.back(), tops);
    if (rc) return rc;
    const long long id = *std::max_element(tops.begin(), tops.end()) + 1;
    std::vector<unsigned long long> bounds(size_t(s.G) + 1);
    HIP_TRY(h, hipMemcpy(bounds.data(), s.d_bounds, bounds.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const float* p = static_cast<const float*>(particle);
    const unsigned long long key = host_key(p, h->center, h->width);
    int owner = 0;
    for (int r = 1; r < s.G; ++r) if (bounds[size_t(r)] <= key) owner = r;
    long long ok = 1;
    if (owner == s.me) {
        if (h->n_local >= size_t(sh.seg_cap)) ok = 0;
        else {
            float4 rec[3] = {make_float4(p[0], p[1], p[2], p[9]), make_float4(p[3], p[4], p[5], 0.f), make_float4(p[6], p[7], p[8], 0.f)};
            const int id32 = int(id);
            const size_t k = h->n_local;
            HIP_TRY(h, hipMemcpyAsync(sh.own_pos() + k, &rec[0], sizeof(float4), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(sh.vel + k, &rec[1], sizeof(float4), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(sh.acc + k, &rec[2], sizeof(float4), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(sh.ids + k, &id32, sizeof(int), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            h->n_local = k + 1;
            rc = push_count(h);
            if (rc) return rc;
        }
    }
    std::vector<long long> oks;   // every rank returns the same verdict
    rc = gather_ll(h, ok, oks);
    if (rc) return rc;
    if (*std::min_element(oks.begin(), oks.end()) == 0) return fail(h, NBODY_ERR_CAPACITY, "spatial shards: the rank that owns the new body's key range is full");
    s.count_delta += 1;
    return NBODY_OK;
}

// Vec::swap_remove(index): `index` counts the world's bodies in the order of their indices in the vector; the body with
// the largest index takes the removed one's place (its index), as the reference's last element does
int remove_point(NbodyHandle* h, size_t index) {
    State& s = *h->let;
    Shard& sh = h->sh;
    std::vector<int> ids, sorted;
    int rc = fetch_ids(h, ids, sorted);
    if (rc) return rc;
    std::vector<long long> all;
    rc = gather_ll(h, (long long)sorted.size(), all);
    if (rc) return rc;
    long long total = 0;
    for (long long c : all) total += c;
    if ((long long)index >= total) return fail(h, NBODY_ERR_INVALID, "swap_remove index out of range");
    rc = gather_ll(h, sorted.empty() ? -1LL : (long long)sorted.back(), all);
    if (rc) return rc;
    const long long top = *std::max_element(all.begin(), all.end());
    // the (index+1)-th smallest id of the world: bisection on the value, one count from every rank per round
    long long lo = 0, hi = top;
    while (lo < hi) {
        const long long mid = lo + (hi - lo) / 2;
        const long long mine = (long long)(std::upper_bound(sorted.begin(), sorted.end(), int(mid)) - sorted.begin());
        rc = gather_ll(h, mine, all);
        if (rc) return rc;
        long long below = 0;
        for (long long c : all) below += c;
        if (below >= (long long)index + 1) hi = mid; else lo = mid + 1;
    }
    const int victim = int(lo);
    // the world's last body takes the victim's index
    if (victim != int(top)) {
        auto it = std::find(ids.begin(), ids.end(), int(top));
        if (it != ids.end()) {
            const size_t at = size_t(it - ids.begin());
            HIP_TRY(h, hipMemcpy(sh.ids + at, &victim, sizeof(int), hipMemcpyHostToDevice));
            ids[at] = victim;   // (if this rank also holds the victim, the search below must find the ORIGINAL: see the index guard)
            auto vit = std::find(ids.begin(), ids.end(), victim);
            if (vit != ids.end() && size_t(vit - ids.begin()) == at) vit = std::find(vit + 1, ids.end(), victim);
            if (vit != ids.end()) {
                const size_t j = size_t(vit - ids.begin()), tail = h->n_local - 1;
                if (j != tail) {
                    HIP_TRY(h, hipMemcpyAsync(sh.own_pos() + j, sh.own_pos() + tail, sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
                    HIP_TRY(h, hipMemcpyAsync(sh.vel + j, sh.vel + tail, sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
                    HIP_TRY(h, hipMemcpyAsync(sh.acc + j, sh.acc + tail, sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
                    HIP_TRY(h, hipMemcpyAsync(sh.ids + j, sh.ids + tail, sizeof(int), hipMemcpyDeviceToDevice, h->stream));
                }
                h->n_local = tail;
                rc = push_count(h);
                if (rc) return rc;
            }
            s.count_delta -= 1;
            return NBODY_OK;
        }
    }
    auto vit = std::find(ids.begin(), ids.end(), victim);
    if (vit != ids.end()) {
        const size_t j = size_t(vit - ids.begin()), tail = h->n_local - 1;
        if (j != tail) {
            HIP_TRY(h, hipMemcpyAsync(sh.own_pos() + j, sh.own_pos() + tail, sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(sh.vel + j, sh.vel + tail, sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(sh.acc + j, sh.acc + tail, sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(sh.ids + j, sh.ids + tail, sizeof(int), hipMemcpyDeviceToDevice, h->stream));
        }
        h->n_local = tail;
        rc = push_count(h);
        if (rc) return rc;
    }
    s.count_delta -= 1;
    return NBODY_OK;
}

int stats(NbodyHandle* h, NbodyLetStats* out) {
    State& s = *h->let;
    if (h->profiling && (s.ev_live[0] || s.ev_live[1])) {   // the last pass's phase events: wait for them
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        collect_phase_times(s, s.ev_set);
        collect_phase_times(s, s.ev_set ^ 1);
    }
    s.st.host_syncs = s.host_syncs;
    s.st.migrant_respills = s.spills;
    s.st.node_array_peak_bytes = s.node_array_peak * sizeof(LetRecord);
    // every buffer of node records this rank has: its slice, the array the walk runs over, the export lists, the staged imports
    s.st.node_array_bytes = uint64_t(s.held_cap + size_t(std::max(0, s.local_cap)) + s.let_send_cap + s.let_recv_cap) * sizeof(LetRecord);
    *out = s.st;
    return NBODY_OK;
}

int reset_stats(NbodyHandle* h) {
    State& s = *h->let;
    s.st = NbodyLetStats{};
    s.host_syncs = 0; s.spills = 0; s.node_array_peak = 0; s.migrated_seen = 0;
    s.ev_live[0] = s.ev_live[1] = false;
    HIP_TRY(h, hipMemsetAsync(s.d_flags + 2, 0, sizeof(int), h->stream));
    return NBODY_OK;
}

// bookkeeping after a pass, from the numbers of the step's report block (rep = h_pin after the copy of d_report)
static void account_from(NbodyHandle* h, State& s, const int* rep) {
    const int G = s.G, gg = G * G;
    const int* let_m = rep;
    const int* offsets = rep + 2 * gg;
    const int* tree_info = rep + 2 * gg + G + 1;
    const int* flags = rep + 2 * gg + G + 4;
    uint64_t sent = 0, rec = 0;
    if (h->comm_ready)
        for (int r = 0; r < G; ++r) {
            if (r == s.me) continue;
            sent += uint64_t(std::max(0, let_m[s.me * G + r]));
            rec += uint64_t(std::max(0, let_m[r * G + s.me]));
        }
    s.st.nodes_sent += sent;
    s.st.nodes_received += rec;
    s.st.nodes_local += uint64_t(std::max(0, tree_info[0]));
    s.st.nodes_global += uint64_t(std::max(0, offsets[G]));
    s.node_array_peak = std::max<uint64_t>(s.node_array_peak, uint64_t(std::max(0, tree_info[0])) + rec);
    h->stats.tree_nodes = uint64_t(std::max(1, offsets[G]));   // (an empty world: the reference's empty root, barnes_hut.rs:145)
    h->n_local = size_t(std::max(0, tree_info[2]));            // the live own count (bodies in the local build)
    h->seg_count_host[0] = int(h->n_local);
    h->count_dirty = false;
    const uint64_t partners = uint64_t(std::max(0, G - 1));
    const uint64_t migrated_now = uint64_t(std::max(0, flags[2]));
    s.st.bytes_sent += sent * sizeof(LetRecord) + (migrated_now - std::min(migrated_now, s.migrated_seen)) * sizeof(Migrant) +
                       partners * (2 * uint64_t(G) * sizeof(int) + sizeof(EndInfo) + sizeof(RoundB));
    s.migrated_seen = migrated_now;
    s.st.bytes_allgather_equivalent += partners * uint64_t(std::max(0, tree_info[2])) * 16ull;
    s.st.bodies_migrated = migrated_now;
}

// the emulation's bookkeeping (debug_phase): the same numbers, fetched with a synchronisation of their own
static int account(NbodyHandle* h, State& s) {
    launch_report(h->stream, s.d_let_matrix, s.d_mig_matrix, s.d_offsets, s.d_tree_info, s.d_flags, s.G, s.d_report);
    HIP_TRY(h, hipMemcpyAsync(s.h_pin, s.d_report, sizeof(int) * report_ints(s.G), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(s.h_pin + report_ints(s.G), s.d_let_count, sizeof(int) * s.G, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const bool was = h->comm_ready;
    h->comm_ready = false;   // (the emulation counts what debug_exchange moved: recv_n; exports from the rank's own counts)
    account_from(h, s, s.h_pin);
    h->comm_ready = was;
    uint64_t sent = 0, rec = 0;
    for (int r = 0; r < s.G; ++r) { sent += uint64_t(std::max(0, s.h_pin[report_ints(s.G) + r])); rec += uint64_t(s.recv_n[r]); }
    s.st.nodes_sent += sent;
    s.st.nodes_received += rec;
    s.st.bytes_sent += sent * sizeof(LetRecord);
    s.node_array_peak = std::max<uint64_t>(s.node_array_peak, uint64_t(std::max(0, s.h_pin[2 * s.G * s.G + s.G + 1])) + rec);
    if (h->profiling) collect_phase_times(s, s.ev_set);
    s.count_delta = 0;
    return NBODY_OK;
}

// Where the variable-size rounds put their messages (host arithmetic shared by sender and receiver: both sides of a
// pair read the same all-gathered matrix, so the byte counts of a send and of the receive that meets it are the same
// expression).  m = the G x G count matrix, row r = what rank r sends to everybody; `clamp` = the most a sender's list holds.
// out_at[r] / in_at[r]: record offsets of the message to / from rank r in the sender's packed buffer / the receive buffer;
// n_out[r] / n_in[r]: records; returns the total received.
size_t exchange_layout(const int* m, int G, int me, long long clamp, bool packed_send, size_t send_stride, size_t* out_at, size_t* n_out,
                       size_t* in_at, size_t* n_in) {
    size_t out_run = 0, in_run = 0;
    for (int r = 0; r < G; ++r) {
        const long long out = r == me ? 0 : std::min<long long>(std::max(0, m[me * G + r]), clamp);
        const long long in = r == me ? 0 : std::min<long long>(std::max(0, m[r * G + me]), clamp);
        n_out[r] = size_t(out);
        n_in[r] = size_t(in);
        out_at[r] = packed_send ? out_run : size_t(r) * send_stride;
        in_at[r] = in_run;
        out_run += size_t(out);
        in_run += size_t(in);
    }
    return in_run;
}

static int flags_to_error(NbodyHandle* h, State& s, int f) {
    if (!f) return NBODY_OK;
    const std::string who = " (flag word " + std::to_string(f) + ")";
    if (f & kFlagBigGroup) return fail(h, NBODY_ERR_TREE_DEPTH, "spatial shards: more than 4096 bodies of one rank share 16 levels of the tree (a clump inside a cell 1.5e-5 of the box wide): beyond what the device build sorts, and a spatial rank has no host build to hand the step to" + who);
    if (f & kFlagDeep) return fail(h, NBODY_ERR_TREE_DEPTH, "spatial shards: two bodies separate only below the device build's 42 levels (coincident?)" + who);
    if (f & kFlagCapacity) return fail(h, NBODY_ERR_CAPACITY, "spatial shards: a rank's capacity is exhausted by immigrants");
    if (f & (kFlagNodeCap | kFlagNodeCapLocal)) return fail(h, NBODY_ERR_CAPACITY, "spatial shards: node array too small");
    return fail(h, NBODY_ERR_INVALID, "spatial shards: the device raised flag " + std::to_string(f));
}

// one grouped send/recv round of records of `rec` bytes: to rank r n_out[r] records from send + out_at[r], from rank r n_in[r] into recv + in_at[r]
static int variable_round(NbodyHandle* h, State& s, const char* send, char* recv, size_t rec, const size_t* out_at, const size_t* n_out, const size_t* in_at,
                          const size_t* n_in) {
    TP_TRY(h, h->tp->group_begin());
    for (int r = 0; r < s.G; ++r) {
        if (n_out[r] > 0) TP_TRY(h, h->tp->send(send + out_at[r] * rec, n_out[r] * rec, r, h->stream));
        if (n_in[r] > 0) TP_TRY(h, h->tp->recv(recv + in_at[r] * rec, n_in[r] * rec, r, h->stream));
    }
    TP_TRY(h, h->tp->group_end());
    return NBODY_OK;
}

// exchange 0 with the sizes the host knows exactly (m = the migrant count matrix on the host)
static int migrants_exact(NbodyHandle* h, State& s, const int* m) {
    size_t out_at[kMaxRanks], n_out[kMaxRanks], in_at[kMaxRanks], n_in[kMaxRanks];
    const size_t total_in = exchange_layout(m, s.G, s.me, (long long)h->sh.seg_cap, true, 0, out_at, n_out, in_at, n_in);
    int rc = ensure_mig_recv(h, s, total_in, 0);
    if (rc) return rc;
    rc = variable_round(h, s, reinterpret_cast<const char*>(s.d_send_mig), reinterpret_cast<char*>(s.d_recv_mig), sizeof(Migrant), out_at, n_out, in_at, n_in);
    if (rc) return rc;
    s.mig_in = int(total_in);
    return NBODY_OK;
}

// what the next step posts its migrant messages with: twice the last count and a little, the same on every rank
static int draw_prediction(NbodyHandle* h, State& s, const int* mig_m) {
    const int G = s.G;
    size_t out_need = 0, in_need = 0;
    for (int a = 0; a < G; ++a)
        for (int b = 0; b < G; ++b) {
            const int p = a == b ? 0 : int(std::min<long long>(2LL * std::max(0, mig_m[a * G + b]) + 32, (long long)h->sh.seg_cap));
            s.h_pred[size_t(a) * G + b] = p;
            if (a == s.me) out_need += size_t(p);
            if (b == s.me) in_need += size_t(p);
        }
    // (the host is synchronised here: growing a buffer costs nothing but the call)
    if (out_need > s.slot_out_cap) {
        if (s.d_slot_out) (void)hipFree(s.d_slot_out);
        s.d_slot_out = nullptr; s.slot_out_cap = 0;
        const size_t cap = out_need + out_need / 2 + 1024;
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&s.d_slot_out), cap * sizeof(Migrant)));
        s.slot_out_cap = cap;
    }
    int rc = ensure_mig_recv(h, s, in_need, 0);
    if (rc) return rc;
    std::memcpy(s.h_pred_pin, s.h_pred.data(), sizeof(int) * size_t(G) * G);
    HIP_TRY(h, hipMemcpyAsync(s.d_pred, s.h_pred_pin, sizeof(int) * size_t(G) * G, hipMemcpyHostToDevice, h->stream));   // (read by the NEXT pass; the staging is rewritten only after that pass's own synchronisation)
    s.have_pred = true;
    return NBODY_OK;
}

// ---- production: the exchanges through the handle's transport (RCCL; or the one-device transport), on the handle's stream.
// ONE host synchronisation per pass in the steady state: after phase 3 the host reads one block (k_let_report: both count
// matrices, the slice offsets, the build's numbers, the flags) -- it needs the export counts there to post the node round --
// and does the pass's bookkeeping and error check from it.  The migrant round does not wait for its counts: its messages are
// posted with sizes drawn from the previous step's counts (draw_prediction); if some pair has more migrants than that, every
// rank sees it in the all-gathered matrix (kFlagMigSpill), nothing is committed, and after the synchronisation the round is
// made again with the exact sizes and phases 1-3 are repeated (NbodyLetStats.migrant_respills counts such steps).
static int pass(NbodyHandle* h, float dt, bool is_step) {
    State& s = *h->let;
    const int G = s.G, gg = G * G;
    if (G > 1 && !h->comm_ready) return fail(h, NBODY_ERR_COMM, "world_size > 1 but nbody_comm_init has not been called");
    const bool comm = h->comm_ready;   // (a world of one WITH a communicator still runs every collective: the 1-rank rehearsal)
    size_t out_at[kMaxRanks], n_out[kMaxRanks], in_at[kMaxRanks], n_in[kMaxRanks];
    if (h->profiling) {   // the previous pass's phase events have all completed by the time this pass synchronises; read them now if they have
        const int prev = s.ev_set ^ 1;
        if (s.ev_live[prev] && hipEventQuery(s.ev[prev][9]) == hipSuccess) collect_phase_times(s, prev);
    }
    int rc = phase0(h, s, dt, is_step);
    if (rc) return rc;
    int slots_in = -1;
    if (comm) {   // exchange 0: the counts (row r = what rank r sends to everybody), then the migrants themselves
        HIP_TRY(h, hipMemcpyAsync(s.d_mig_matrix + size_t(s.me) * G, s.d_send_count, sizeof(int) * G, hipMemcpyDeviceToDevice, h->stream));
        TP_TRY(h, h->tp->all_gather(s.d_mig_matrix, sizeof(int) * size_t(G), h->stream));
        if (s.have_pred) {
            launch_spec_check(h->stream, s.d_mig_matrix, s.d_pred, G, s.d_flags);
            launch_slot_migrants(h->stream, s.d_send_mig, int(std::min<size_t>(h->n_local, size_t(h->sh.seg_cap))), s.d_send_off, s.d_pred, G, s.me, s.d_slot_out);
            const size_t total_in = exchange_layout(s.h_pred.data(), G, s.me, (long long)h->sh.seg_cap, true, 0, out_at, n_out, in_at, n_in);
            rc = variable_round(h, s, reinterpret_cast<const char*>(s.d_slot_out), reinterpret_cast<char*>(s.d_recv_mig), sizeof(Migrant), out_at, n_out, in_at, n_in);
            if (rc) return rc;
            slots_in = int(total_in);
            s.mig_in = int(total_in);   // (the host's bound of what may arrive)
        } else {
            HIP_TRY(h, hipMemcpyAsync(s.h_pin, s.d_mig_matrix, sizeof(int) * gg, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            s.host_syncs += 1;
            rc = migrants_exact(h, s, s.h_pin);
            if (rc) return rc;
        }
    }
    const size_t n_before = h->n_local;   // (phase 1 raises the host's bound by what may arrive: a repeated round starts from here again)
    const int* rep = s.h_pin;
    for (int attempt = 0;; ++attempt) {
        rc = phase1(h, s, slots_in);
        if (rc) return rc;
        if (comm) TP_TRY(h, h->tp->all_gather(s.d_ends, sizeof(EndInfo), h->stream));   // exchange 1
        rc = phase2(h, s);
        if (rc) return rc;
        if (comm) TP_TRY(h, h->tp->all_gather(s.d_rb, sizeof(RoundB), h->stream));      // exchange 2
        rc = phase3(h, s);
        if (rc) return rc;
        if (comm) {   // exchange 3, first half: the export counts (row r of the matrix = what rank r sends to everybody)
            HIP_TRY(h, hipMemcpyAsync(s.d_let_matrix + size_t(s.me) * G, s.d_let_count, sizeof(int) * G, hipMemcpyDeviceToDevice, h->stream));
            TP_TRY(h, h->tp->all_gather(s.d_let_matrix, sizeof(int) * size_t(G), h->stream));
        }
        launch_report(h->stream, s.d_let_matrix, s.d_mig_matrix, s.d_offsets, s.d_tree_info, s.d_flags, G, s.d_report);
        HIP_TRY(h, hipMemcpyAsync(s.h_pin, s.d_report, sizeof(int) * report_ints(G), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));   // the one synchronisation of the pass
        s.host_syncs += 1;
        const int f = rep[2 * gg + G + 4];
        if (!(f & kFlagMigSpill)) break;
        // more migrants than their messages were posted for: every rank is here (they all saw the same matrices).  Nothing was
        // committed; the emigrants are still packed in d_send_mig: the round again, exactly, then phases 1-3 again.
        if ((f & ~kFlagMigSpill) || attempt > 0) return flags_to_error(h, s, (f & ~kFlagMigSpill) ? (f & ~kFlagMigSpill) : f);
        s.spills += 1;
        HIP_TRY(h, hipMemsetAsync(s.d_flags, 0, sizeof(int), h->stream));
        h->n_local = n_before;
        std::vector<int> mig(rep + gg, rep + 2 * gg);   // (h_pin is about to be reused)
        rc = migrants_exact(h, s, mig.data());
        if (rc) return rc;
        slots_in = -1;
    }
    {
        const int f = rep[2 * gg + G + 4];
        if (f) return flags_to_error(h, s, f);
    }
    if (h->tp) { rc = h->tp->check(); if (rc) return fail(h, rc, h->tp->error()); }
    account_from(h, s, rep);
    std::fill(s.recv_n.begin(), s.recv_n.end(), 0);
    if (comm) {
        const size_t total_in = exchange_layout(rep, G, s.me, kNoClamp, true, 0, out_at, n_out, in_at, n_in);
        {   // my lists lie one after the other; a buffer that was too small for them is grown and written again (the counts
            // and every node's partners are still on the device: only the last kernel of the export runs twice)
            size_t total_out = 0;
            for (int r = 0; r < G; ++r) total_out += n_out[r];
            rc = ensure_let_send(h, s, total_out);
            if (rc) return rc;
        }
        for (int r = 0; r < G; ++r) s.recv_n[r] = int(n_in[r]);
        rc = ensure_let_recv(h, s, total_in, 0);   // (sized for a slice's worth at creation: only a rank that imports more than it can own grows it)
        if (rc) return rc;
        rc = draw_prediction(h, s, rep + gg);   // (before h_pin is reused; uploads next step's sizes)
        if (rc) return rc;
        rc = variable_round(h, s, reinterpret_cast<const char*>(s.d_let_send), reinterpret_cast<char*>(s.d_let_recv), sizeof(LetRecord), out_at, n_out, in_at, n_in);
        if (rc) return rc;
    }
    rc = phase4(h, s, dt, is_step);
    if (rc) return rc;
    s.ev_set ^= 1;   // (the next pass records into the other set; this one's are read once they have completed)
    s.count_delta = 0;   // (the table count_global reads is this pass's)
    return NBODY_OK;
}

int step(NbodyHandle* h, float dt) { return pass(h, dt, true); }

int update_forces(NbodyHandle* h) { return pass(h, 0.f, false); }

// ---- one-process emulation: phases and exchanges driven from outside
int debug_phase(NbodyHandle* h, int phase, float dt) {
    State& s = *h->let;
    switch (phase) {
        case 0: return phase0(h, s, dt, true);
        case 10: return phase0(h, s, dt, false);   // the same for a force pass outside a step (no drift)
        case 1: return phase1(h, s);
        case 2: return phase2(h, s);
        case 3: {
            std::fill(s.recv_n.begin(), s.recv_n.end(), 0);
            s.staged = 0;
            int rc = phase3(h, s);
            if (rc) return rc;
            HIP_TRY(h, hipMemcpyAsync(s.h_pin, s.d_list_first + s.G, sizeof(int), hipMemcpyDeviceToHost, h->stream));   // all my lists together
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            return ensure_let_send(h, s, size_t(std::max(0, s.h_pin[0])));
        }
        case 4: { int rc = phase4(h, s, dt, true); if (rc) return rc; rc = account(h, s); return rc ? rc : check_flags(h); }
        case 14: { int rc = phase4(h, s, dt, false); if (rc) return rc; rc = account(h, s); return rc ? rc : check_flags(h); }
        default: return fail(h, NBODY_ERR_INVALID, "unknown phase");
    }
}

int debug_exchange(NbodyHandle* h, NbodyHandle* peer, int which) {
    State& s = *h->let;
    State& p = *peer->let;
    const int me = s.me, pr = p.me;
    HIP_TRY(h, hipStreamSynchronize(peer->stream));
    switch (which) {
        case 0: { // migrants: the peer's run for me, behind what has arrived already
            int cnt = 0, off = 0;
            HIP_TRY(h, hipMemcpy(&cnt, p.d_send_count + me, sizeof(int), hipMemcpyDeviceToHost));
            HIP_TRY(h, hipMemcpy(&off, p.d_send_off + me, sizeof(int), hipMemcpyDeviceToHost));
            if (cnt > 0) {
                int rc = ensure_mig_recv(h, s, size_t(s.mig_in) + size_t(cnt), size_t(s.mig_in));
                if (rc) return rc;
                HIP_TRY(h, hipMemcpyAsync(s.d_recv_mig + s.mig_in, p.d_send_mig + off, size_t(cnt) * sizeof(Migrant), hipMemcpyDeviceToDevice,
                                          h->stream));
                s.mig_in += cnt;
            }
            break;
        }
        case 1:
            HIP_TRY(h, hipMemcpyAsync(s.d_ends + pr, p.d_ends + pr, sizeof(EndInfo), hipMemcpyDeviceToDevice, h->stream));
            break;
        case 2:
            HIP_TRY(h, hipMemcpyAsync(s.d_rb + pr, p.d_rb + pr, sizeof(RoundB), hipMemcpyDeviceToDevice, h->stream));
            break;
        case 3: {
            int n = 0;
            HIP_TRY(h, hipMemcpy(&n, p.d_let_count + me, sizeof(int), hipMemcpyDeviceToHost));
            int first = 0;
            HIP_TRY(h, hipMemcpy(&first, p.d_list_first + me, sizeof(int), hipMemcpyDeviceToHost));
            if (size_t(first) + size_t(n) > p.let_send_cap) return fail(h, NBODY_ERR_CAPACITY, "emulation: the peer's export lists were not written whole");
            for (int r = pr; r < s.G; ++r)
                if (r != me && s.recv_n[r] != 0) return fail(h, NBODY_ERR_INVALID, "emulation: the node lists must arrive in rank order");
            {
                int rc = ensure_let_recv(h, s, s.staged + size_t(n), s.staged);
                if (rc) return rc;
            }
            if (n > 0)
                HIP_TRY(h, hipMemcpyAsync(s.d_let_recv + s.staged, p.d_let_send + size_t(first), size_t(n) * sizeof(LetRecord), hipMemcpyDeviceToDevice, h->stream));
            s.staged += size_t(n);
            s.recv_n[pr] = n;
            break;
        }
        default: return fail(h, NBODY_ERR_INVALID, "unknown exchange");
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return NBODY_OK;
}

}  // namespace let
}  // namespace nbody

extern "C" int nbody_debug_let_set_balance(NbodyHandle* h, int by_work) {
    if (!h || !h->let) return NBODY_ERR_INVALID;
    h->let->by_work = by_work != 0;
    return NBODY_OK;
}
extern "C" int nbody_debug_let_set_prune(NbodyHandle* h, int prune) {
    if (!h || !h->let) return NBODY_ERR_INVALID;
    h->let->prune = prune != 0;
    return NBODY_OK;
}

extern "C" int nbody_debug_let_bounds(NbodyHandle* h, unsigned long long* out) {
    if (!h || !h->let || !out) return NBODY_ERR_INVALID;
    if (hipMemcpy(out, h->let->d_bounds, (size_t(h->let->G) + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return NBODY_ERR_HIP;
    return NBODY_OK;
}
