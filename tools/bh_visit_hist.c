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
