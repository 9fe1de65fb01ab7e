// bh_visit_hist.c -- CPU replay of the stackless skip-link walk (kernels_bh.hip k_bh_walk) that counts
// visits per NODE, for tools/bh_visit_hist.py (where do the opening tests land in the tree?).
// gcc -O3 -fopenmp -shared -fPIC -ffp-contract=off
#include <stdint.h>
#include <stddef.h>
void walk_counts(const float* com_mass, const float* width, const int32_t* skip, int n_nodes,
                 const float* pos4, int n, float theta2, uint32_t* visits, uint32_t* accepts) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int b = 0; b < n; ++b) {
        const float px = pos4[4 * b], py = pos4[4 * b + 1], pz = pos4[4 * b + 2];
        int i = 0;
        while (i < n_nodes) {
            const float rx = com_mass[4 * i] - px, ry = com_mass[4 * i + 1] - py, rz = com_mass[4 * i + 2] - pz;
            const float r2 = (rx * rx + ry * ry) + rz * rz;
            __atomic_fetch_add(&visits[i], 1u, __ATOMIC_RELAXED);
            if (width[i] * width[i] < theta2 * r2) { __atomic_fetch_add(&accepts[i], 1u, __ATOMIC_RELAXED); i = skip[i]; }
            else i = i + 1;
        }
    }
}

// per group of `gsz` consecutive tree-order bodies: how many distinct nodes the group's walks visit (the
// iterations of a wave-cooperative walk), and the sum of the members' own visits
void union_counts(const float* com_mass, const float* width, const int32_t* skip, int n_nodes, const float* pos4,
                  const int32_t* order, int n, float theta2, int gsz, int64_t* out_union, int64_t* out_sum, int64_t* out_max) {
    int64_t tot_u = 0, tot_s = 0, tot_m = 0;
    const int n_groups = (n + gsz - 1) / gsz;
#pragma omp parallel reduction(+ : tot_u, tot_s, tot_m)
    {
        int32_t* stamp = (int32_t*)__builtin_malloc(sizeof(int32_t) * (size_t)n_nodes);
        for (int i = 0; i < n_nodes; ++i) stamp[i] = -1;
#pragma omp for schedule(dynamic, 4)
        for (int gi = 0; gi < n_groups; ++gi) {
            int64_t u = 0, mx = 0;
            for (int q = gi * gsz; q < (gi + 1) * gsz && q < n; ++q) {
                const int b = order[q];
                const float px = pos4[4 * b], py = pos4[4 * b + 1], pz = pos4[4 * b + 2];
                int i = 0;
                int64_t mine = 0;
                while (i < n_nodes) {
                    const float rx = com_mass[4 * i] - px, ry = com_mass[4 * i + 1] - py, rz = com_mass[4 * i + 2] - pz;
                    const float r2 = (rx * rx + ry * ry) + rz * rz;
                    if (stamp[i] != gi) { stamp[i] = gi; ++u; }
                    ++mine;
                    if (width[i] * width[i] < theta2 * r2) i = skip[i]; else i = i + 1;
                }
                tot_s += mine;
                if (mine > mx) mx = mine;
            }
            tot_u += u;
            tot_m += mx;
        }
        __builtin_free(stamp);
    }
    *out_union = tot_u; *out_sum = tot_s; *out_max = tot_m;
}

// Simulation of a wave-cooperative walk with a window of W consecutive node records staged per wave:
// wave-uniform node index i; lane l takes part iff i >= resume[l]; next = i + 1 if any taking-part lane opens the
// node, else max(skip(i), min over lanes of resume).  Counts iterations, iterations in which no lane takes part,
// and window reloads (i outside [base, base + W)).
void coop_counts(const float* com_mass, const float* width, const int32_t* skip, int n_nodes, const float* pos4,
                 const int32_t* order, int n, float theta2, int gsz, int W, int K, int use_min_resume,
                 int64_t* out_iters, int64_t* out_dead, int64_t* out_reloads, int64_t* out_max_iters) {
    int64_t it = 0, dead = 0, rel = 0, mx = 0;
    const int n_groups = (n + gsz - 1) / gsz;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : it, dead, rel) reduction(max : mx)
    for (int gi = 0; gi < n_groups; ++gi) {
        for (int k = 0; k < K; ++k) {
            const int s0 = (int)((long long)n_nodes * k / K), s1 = (int)((long long)n_nodes * (k + 1) / K);
            int resume[256];
            int cnt = 0;
            for (int q = gi * gsz; q < (gi + 1) * gsz && q < n; ++q, ++cnt) {
                // entry: replay the walk from the root up to the first visited node >= s0 (what walk_entry computes)
                const int b = order[q];
                const float px = pos4[4 * b], py = pos4[4 * b + 1], pz = pos4[4 * b + 2];
                int i = 0;
                while (i < s0) {
                    const float rx = com_mass[4 * i] - px, ry = com_mass[4 * i + 1] - py, rz = com_mass[4 * i + 2] - pz;
                    const float r2 = (rx * rx + ry * ry) + rz * rz;
                    if (width[i] * width[i] < theta2 * r2) i = skip[i];
                    else if (skip[i] <= s0) i = i + 1 > skip[i] ? skip[i] : i + 1;  // (descend; a subtree ending before s0 is walked through)
                    else i = i + 1;
                }
                resume[cnt] = i;
            }
            int i = s0, base = -1000000;
            int64_t my = 0;
            while (i < s1) {
                if (i < base || i >= base + W) { base = i; ++rel; }
                int any_active = 0, any_open = 0, min_res = 0x7fffffff;
                for (int l = 0; l < cnt; ++l) {
                    if (i >= resume[l]) {
                        any_active = 1;
                        const int b = order[gi * gsz + l];
                        const float rx = com_mass[4 * i] - pos4[4 * b], ry = com_mass[4 * i + 1] - pos4[4 * b + 1], rz = com_mass[4 * i + 2] - pos4[4 * b + 2];
                        const float r2 = (rx * rx + ry * ry) + rz * rz;
                        if (width[i] * width[i] < theta2 * r2) resume[l] = skip[i];
                        else { any_open = 1; resume[l] = i + 1; }
                    }
                    if (resume[l] < min_res) min_res = resume[l];
                }
                ++my;
                if (!any_active) ++dead;
                int nxt = any_open ? i + 1 : skip[i];
                if (use_min_resume && !any_open && min_res > nxt) nxt = min_res;
                i = nxt;
            }
            it += my;
            if (my > mx) mx = my;
        }
    }
    *out_iters = it; *out_dead = dead; *out_reloads = rel; *out_max_iters = mx;
}

// Items of a body group under the "fine near the diagonal, doubling runs away from it" split of the node range:
// Kf fine segments of equal node count; the group's own place d = g * Kf / n_groups; fine segments within R of d are
// items of their own, beyond that runs of 2, 4, 8, ... fine segments.  Writes the item boundaries (fine indices) to
// `cuts` (at most max_cuts), returns the number of items.
static int g_run_cap = 1 << 30;
void set_run_cap(int c) { g_run_cap = c; }
static int make_items(int g, int n_groups, int Kf, int R, int* cuts, int max_cuts) {
    const int d = (int)((long long)g * Kf / n_groups);
    int lo = d - R < 0 ? 0 : d - R, hi = d + R + 1 > Kf ? Kf : d + R + 1;
    int left[512], nl = 0;
    { int e = lo, len = 2; while (e > 0) { int s = e - len < 0 ? 0 : e - len; left[nl++] = s; e = s; len *= 2; if (len > g_run_cap) len = g_run_cap; } }
    int n = 0;
    for (int q = nl - 1; q >= 0; --q) cuts[n++] = left[q];
    for (int f = lo; f < hi; ++f) cuts[n++] = f;
    { int s = hi, len = 2; while (s < Kf) { cuts[n++] = s; s = s + len > Kf ? Kf : s + len; len *= 2; if (len > g_run_cap) len = g_run_cap; } }
    cuts[n] = Kf;
    (void)max_cuts;
    return n;
}

void coop_items(const float* com_mass, const float* width, const int32_t* skip, int n_nodes, const float* pos4,
                const int32_t* order, int n, float theta2, int Kf, int R, int64_t* out_items, int64_t* out_iters,
                int64_t* out_max, int32_t* hist /* [64]: items by iterations / 64 */) {
    const int gsz = 64;
    const int n_groups = (n + gsz - 1) / gsz;
    int64_t items = 0, iters = 0, mx = 0;
    for (int q = 0; q < 64; ++q) hist[q] = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : items, iters) reduction(max : mx)
    for (int gi = 0; gi < n_groups; ++gi) {
        int cuts[1024];
        const int ni = make_items(gi, n_groups, Kf, R, cuts, 1023);
        for (int k = 0; k < ni; ++k) {
            const int s0 = (int)((long long)n_nodes * cuts[k] / Kf), s1 = (int)((long long)n_nodes * cuts[k + 1] / Kf);
            int resume[64];
            int cnt = 0, first = 0x7fffffff;
            for (int q = gi * gsz; q < (gi + 1) * gsz && q < n; ++q, ++cnt) {
                const int b = order[q];
                const float px = pos4[4 * b], py = pos4[4 * b + 1], pz = pos4[4 * b + 2];
                int i = 0;
                while (i < s0) {
                    const float rx = com_mass[4 * i] - px, ry = com_mass[4 * i + 1] - py, rz = com_mass[4 * i + 2] - pz;
                    const float r2 = (rx * rx + ry * ry) + rz * rz;
                    if (width[i] * width[i] < theta2 * r2) i = skip[i]; else i = i + 1;
                }
                resume[cnt] = i;
                if (i < first) first = i;
            }
            int i = first;
            int64_t my = 0;
            while (i < s1) {
                int any_open = 0;
                for (int l = 0; l < cnt; ++l) {
                    if (i >= resume[l]) {
                        const int b = order[gi * gsz + l];
                        const float rx = com_mass[4 * i] - pos4[4 * b], ry = com_mass[4 * i + 1] - pos4[4 * b + 1], rz = com_mass[4 * i + 2] - pos4[4 * b + 2];
                        const float r2 = (rx * rx + ry * ry) + rz * rz;
                        if (width[i] * width[i] < theta2 * r2) resume[l] = skip[i];
                        else { any_open = 1; resume[l] = i + 1; }
                    }
                }
                ++my;
                i = any_open ? i + 1 : skip[i];
            }
            ++items;
            iters += my;
            if (my > mx) mx = my;
            const int hb = my / 64 > 63 ? 63 : (int)(my / 64);
#pragma omp atomic
            hist[hb]++;
        }
    }
    *out_items = items; *out_iters = iters; *out_max = mx;
}

// Window policy of the cooperative walk: fills and bytes per group for a fixed window (wmin == wmax) or an
// adaptive one: after leaving a window in which `used` records were visited, the next is 2x as long if
// used * grow_div >= W, half as long if used * shrink_div < W (clamped to [wmin, wmax]).  K = 1 (whole tree per group).
void coop_window_policy(const float* com_mass, const float* width, const int32_t* skip, int n_nodes, const float* pos4,
                        const int32_t* order, int n, float theta2, int wmin, int wmax, int grow_div, int shrink_div,
                        int64_t* out_iters, int64_t* out_fills, int64_t* out_records) {
    const int gsz = 64;
    const int n_groups = (n + gsz - 1) / gsz;
    int64_t it = 0, fills = 0, recs = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : it, fills, recs)
    for (int gi = 0; gi < n_groups; ++gi) {
        int resume[64];
        int cnt = 0;
        for (int q = gi * gsz; q < (gi + 1) * gsz && q < n; ++q, ++cnt) resume[cnt] = 0;
        int i = 0, base = 0, W = wmin, used = 0;
        fills = fills + 1; recs += W;
        while (i < n_nodes) {
            if (i >= base + W) {
                if (used * grow_div >= W) W = W * 2 > wmax ? wmax : W * 2;
                else if (used * shrink_div < W) W = W / 2 < wmin ? wmin : W / 2;
                base = i; used = 0;
                ++fills; recs += W;
            }
            ++used;
            int any_open = 0;
            for (int l = 0; l < cnt; ++l) {
                if (i >= resume[l]) {
                    const int b = order[gi * gsz + l];
                    const float rx = com_mass[4 * i] - pos4[4 * b], ry = com_mass[4 * i + 1] - pos4[4 * b + 1], rz = com_mass[4 * i + 2] - pos4[4 * b + 2];
                    const float r2 = (rx * rx + ry * ry) + rz * rz;
                    if (width[i] * width[i] < theta2 * r2) resume[l] = skip[i];
                    else { any_open = 1; resume[l] = i + 1; }
                }
            }
            ++it;
            i = any_open ? i + 1 : skip[i];
        }
    }
    *out_iters = it; *out_fills = fills; *out_records = recs;
}

// Block walk: children of a node stored contiguously; a group (64 bodies) pops a block, tests every child for the
// lanes that opened the parent, pushes the blocks of opened children.  Counts per group: blocks fetched, children
// tested (= the union of visited nodes), maximum stack entries.
void block_walk_counts(const float* com_mass, const float* width, const int32_t* skip, int n_nodes, const float* pos4,
                       const int32_t* order, int n, float theta2, int64_t* out_blocks, int64_t* out_children,
                       int64_t* out_max_stack, int64_t* out_lane_tests) {
    const int gsz = 64;
    const int n_groups = (n + gsz - 1) / gsz;
    int64_t blocks = 0, children = 0, mxs = 0, lt = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : blocks, children, lt) reduction(max : mxs)
    for (int gi = 0; gi < n_groups; ++gi) {
        struct { int node; uint64_t mask; } st[512];
        int sp = 0;
        int cnt = 0;
        for (int q = gi * gsz; q < (gi + 1) * gsz && q < n; ++q) ++cnt;
        const uint64_t all = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
        // the root is a block of one
        st[sp].node = -1; st[sp].mask = all; ++sp;
        while (sp > 0) {
            --sp;
            const int parent = st[sp].node;
            const uint64_t M = st[sp].mask;
            ++blocks;
            int c = parent < 0 ? 0 : parent + 1;
            const int end = parent < 0 ? 1 : skip[parent];
            int kids[8]; uint64_t km[8]; int nk = 0;
            while (c < end) {
                ++children;
                uint64_t mo = 0;
                for (int l = 0; l < cnt; ++l) {
                    if (!((M >> l) & 1)) continue;
                    ++lt;
                    const int b = order[gi * gsz + l];
                    const float rx = com_mass[4 * c] - pos4[4 * b], ry = com_mass[4 * c + 1] - pos4[4 * b + 1], rz = com_mass[4 * c + 2] - pos4[4 * b + 2];
                    const float r2 = (rx * rx + ry * ry) + rz * rz;
                    if (!(width[c] * width[c] < theta2 * r2)) mo |= 1ull << l;
                }
                if (mo && skip[c] > c + 1) { kids[nk] = c; km[nk] = mo; ++nk; }
                c = skip[c];
            }
            for (int q = nk - 1; q >= 0; --q) { st[sp].node = kids[q]; st[sp].mask = km[q]; ++sp; }
            if (sp > mxs) mxs = sp;
        }
    }
    *out_blocks = blocks; *out_children = children; *out_max_stack = mxs; *out_lane_tests = lt;
}
