// nbody_oracle.cpp -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
//
// PARITY UNPINNED: the reference (alxn3/nbody-llm, Rust nightly + un-vendored crates + a
// build.rs that downloads a shader compiler) cannot be built in this container and ships no
// tests, golden vectors or fixtures for this path.  This file is therefore a line-by-line
// restatement of the reference's arithmetic, pinned only by the analytic / hand-computed cases
// in tests/test_oracle_pins.py (two-body orbit, 3-/4-body accelerations, hand-built octrees,
// leaf-drop cases).  Nothing here is shipped or measured as the product: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// What is restated (paths relative to /root/reference):
//   src/shared.rs:61-78    SimulationSettings defaults            -> oracle_default_settings_*
//   src/shared.rs:135-140  LeapFrogIntegrator::integrate_pre_force -> pre_force
//   src/shared.rs:141-148  LeapFrogIntegrator::integrate_after_force -> after_force
//   src/shared.rs:206-229  AABB::contains, Bounds::min/max        -> contains
//   src/shared.rs:236-273  Bounds::new/get_orthant/create_orthant -> Box3
//   src/manual/brute_force.rs:64-82  update_forces                -> bf_update_forces
//   src/manual/brute_force.rs:84-90  step_by                      -> bf_step_by
//   src/manual/barnes_hut.rs:143-183 build_tree                   -> build_tree
//   src/manual/barnes_hut.rs:185-203 calc_force                   -> calc_force
//   src/manual/barnes_hut.rs:250-271 update_forces / step_by      -> bh_update_forces / bh_step_by
//   src/llm/barnes_hut.rs:915-997    calc_force_3d (leaf semantics) -> calc_force_direct (leaf_mode 1)
//
// nalgebra 0.33.2 (Cargo.lock) is not in the container.  Its published algorithm for a
// 3-vector `norm_squared()` is dotc(self,self) with the fixed-size-3 special case
// `a0*b0 + a1*b1 + a2*b2` evaluated left to right, i.e. (x*x + y*y) + z*z with no FMA (rustc
// never contracts).  `norm_sq3` below is the single place that encodes this.
// Vector ops (`r * s`, `v += w`, `v / s`) are component-wise; `Sum` folds left from zero.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared (see oracle/Makefile).  The
// -ffp-contract=off flag is load-bearing: the restatement must round every product.

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <array>
#include <memory>
#include <thread>
#include <future>
#include <algorithm>

namespace {

// PointParticle<F,3>, #[repr(C)] (src/shared.rs:151-158): 10 scalars, 40 B (f32) / 80 B (f64).
template <class F>
struct Body {
    F pos[3];
    F vel[3];
    F acc[3];
    F mass;
};

template <class F>
struct Settings {  // src/shared.rs:61-66, field order g, g_soft, dt, theta2
    F g, g_soft, dt, theta2;
};

// Bounds<F,3> (src/shared.rs:216-273)
template <class F>
struct Box3 {
    F center[3];
    F half_width;
    F width;

    static Box3 make(const F c[3], F width) {  // Bounds::new, shared.rs:236-243
        Box3 b;
        b.center[0] = c[0]; b.center[1] = c[1]; b.center[2] = c[2];
        b.half_width = width * F(0.5);
        b.width = width;
        return b;
    }
    // AABB::contains (shared.rs:210-212): position >= min && position <= max, component-wise
    // "all" semantics of nalgebra's PartialOrd; min/max = center.add_scalar(-/+half_width)
    // (shared.rs:223-229).  NaN compares false -> dropped.
    bool contains(const F p[3]) const {
        for (int i = 0; i < 3; ++i) {
            F lo = center[i] + (-half_width);
            F hi = center[i] + half_width;
            if (!(p[i] >= lo)) return false;
            if (!(p[i] <= hi)) return false;
        }
        return true;
    }
    int get_orthant(const F p[3]) const {  // shared.rs:245-254
        int o = 0;
        for (int i = 0; i < 3; ++i)
            if (p[i] > center[i]) o |= 1 << i;
        return o;
    }
    Box3 create_orthant(int orthant) const {  // shared.rs:256-272
        Box3 b;
        F w = width * F(0.5);
        F hw = half_width * F(0.5);
        for (int i = 0; i < 3; ++i) {
            if (orthant & (1 << i)) b.center[i] = center[i] + hw;
            else                    b.center[i] = center[i] - hw;
        }
        b.half_width = hw;
        b.width = w;
        return b;
    }
};

template <class F>
inline F norm_sq3(const F r[3]) {  // nalgebra dotc, 3-vector special case (see header)
    return (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];
}

// ---------------------------------------------------------------- integrator (DKD leapfrog)
template <class F>
void pre_force(Body<F>* b, size_t n, F dt) {  // shared.rs:135-140
    for (size_t k = 0; k < n; ++k)
        for (int i = 0; i < 3; ++i)
            b[k].pos[i] += (b[k].vel[i] * F(0.5)) * dt;
}

template <class F>
void after_force(Body<F>* b, size_t n, F dt) {  // shared.rs:141-148
    for (size_t k = 0; k < n; ++k) {
        for (int i = 0; i < 3; ++i) b[k].vel[i] += b[k].acc[i] * dt;
        for (int i = 0; i < 3; ++i) b[k].pos[i] += (b[k].vel[i] * F(0.5)) * dt;
    }
}

template <class F>
size_t retain_in_bounds(Body<F>* b, size_t n, const Box3<F>& box) {  // Vec::retain, order kept
    size_t w = 0;
    for (size_t k = 0; k < n; ++k) {
        if (box.contains(b[k].pos)) {
            if (w != k) b[w] = b[k];
            ++w;
        }
    }
    return w;
}

// ---------------------------------------------------------------- brute force
template <class F>
void bf_update_forces(Body<F>* p, size_t n, const Settings<F>& s) {  // brute_force.rs:64-82
    for (size_t k = 0; k < n; ++k) p[k].acc[0] = p[k].acc[1] = p[k].acc[2] = F(0);
    const F g_soft2 = s.g_soft * s.g_soft;
    for (size_t i = 0; i < n; ++i) {
        for (size_t j = 0; j < i; ++j) {
            F r[3] = {p[i].pos[0] - p[j].pos[0], p[i].pos[1] - p[j].pos[1], p[i].pos[2] - p[j].pos[2]};
            F r_dist = std::sqrt(norm_sq3(r) + g_soft2);
            F r_cubed = r_dist * r_dist * r_dist;
            F m_i = p[i].mass, m_j = p[j].mass;
            F force = s.g / r_cubed;
            for (int c = 0; c < 3; ++c) p[i].acc[c] -= (r[c] * force) * m_j;
            for (int c = 0; c < 3; ++c) p[j].acc[c] += (r[c] * force) * m_i;
        }
    }
}

// Row-wise form of the same sum: body k meets its partners in ascending index order, first in
// the i-role (j<k: a_k -= ((p_k-p_j)*f)*m_j) then in the j-role (i>k: a_k += ((p_i-p_k)*f)*m_i).
// IEEE negation is exact, so this is bit-identical to bf_update_forces (checked in
// tests/test_oracle_pins.py) at twice the pair evaluations; it is what a one-thread-per-body
// device kernel computes, and it threads trivially (reported "for context" next to the serial
// reference loop in bench.py).
template <class F>
void bf_update_forces_rows(Body<F>* p, size_t n, const Settings<F>& s, int threads, size_t row0 = 0,
                           size_t row1 = size_t(-1)) {
    if (row1 > n) row1 = n;
    if (row0 > row1) row0 = row1;
    const F g_soft2 = s.g_soft * s.g_soft;
    auto rows = [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k) {
            F a[3] = {F(0), F(0), F(0)};
            for (size_t j = 0; j < k; ++j) {
                F r[3] = {p[k].pos[0] - p[j].pos[0], p[k].pos[1] - p[j].pos[1], p[k].pos[2] - p[j].pos[2]};
                F r_dist = std::sqrt(norm_sq3(r) + g_soft2);
                F force = s.g / (r_dist * r_dist * r_dist);
                for (int c = 0; c < 3; ++c) a[c] -= (r[c] * force) * p[j].mass;
            }
            for (size_t i = k + 1; i < n; ++i) {
                F r[3] = {p[i].pos[0] - p[k].pos[0], p[i].pos[1] - p[k].pos[1], p[i].pos[2] - p[k].pos[2]};
                F r_dist = std::sqrt(norm_sq3(r) + g_soft2);
                F force = s.g / (r_dist * r_dist * r_dist);
                for (int c = 0; c < 3; ++c) a[c] += (r[c] * force) * p[i].mass;
            }
            // accelerations are written after all reads of positions/masses; acc is not an input
            p[k].acc[0] = a[0]; p[k].acc[1] = a[1]; p[k].acc[2] = a[2];
        }
    };
    const size_t m = row1 - row0;
    if (threads <= 1 || m < 64) { rows(row0, row1); return; }
    // every row costs n-1 pairs, so contiguous chunks balance
    std::vector<std::thread> pool;
    size_t chunk = (m + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        size_t k0 = std::min(row1, row0 + t * chunk), k1 = std::min(row1, k0 + chunk);
        if (k0 < k1) pool.emplace_back(rows, k0, k1);
    }
    for (auto& th : pool) th.join();
}

template <class F>
size_t bf_step_by(Body<F>* p, size_t n, const Settings<F>& s, const Box3<F>& box, F dt) {
    pre_force(p, n, dt);                 // brute_force.rs:85
    n = retain_in_bounds(p, n, box);     // :86
    bf_update_forces(p, n, s);           // :87
    after_force(p, n, dt);               // :88
    return n;                            // elapsed += dt is the caller's (:89)
}

// ---------------------------------------------------------------- Barnes-Hut
// OrthNode (barnes_hut.rs:11-35).  Children are heap boxes like the reference's Option<Box<_>>.
template <class F>
struct OrthNode {
    F com[3] = {F(0), F(0), F(0)};
    Box3<F> bounds;
    F mass = F(0);
    std::array<std::unique_ptr<OrthNode>, 8> children;
    const Body<F>* leaf_body = nullptr;  // bookkeeping only (not in the reference): set for 1-body nodes
};

constexpr int kMaxDepth = 192;  // the reference has no guard and overflows its stack on coincident
                                // bodies; the restatement reports an error instead (rc = -2)

template <class F>
std::unique_ptr<OrthNode<F>> build_tree(const std::vector<const Body<F>*>& pts, const Box3<F>& bounds,
                                        int depth, int par_levels, bool* too_deep) {
    auto node = std::make_unique<OrthNode<F>>();
    node->bounds = bounds;
    if (pts.size() == 0) return node;                 // barnes_hut.rs:145
    if (pts.size() == 1) {                            // :146-152
        node->com[0] = pts[0]->pos[0]; node->com[1] = pts[0]->pos[1]; node->com[2] = pts[0]->pos[2];
        node->mass = pts[0]->mass;
        node->leaf_body = pts[0];
        return node;
    }
    if (depth >= kMaxDepth) { *too_deep = true; return node; }
    std::array<std::vector<const Body<F>*>, 8> orth;  // :154-158
    for (const Body<F>* p : pts) orth[bounds.get_orthant(p->pos)].push_back(p);

    if (par_levels > 0) {                             // rayon par_iter over the orthants, :160-170
        std::array<std::future<std::unique_ptr<OrthNode<F>>>, 8> fut;
        for (int i = 0; i < 8; ++i)
            if (!orth[i].empty())
                fut[i] = std::async(std::launch::async, [&, i] {
                    return build_tree<F>(orth[i], bounds.create_orthant(i), depth + 1, par_levels - 1, too_deep);
                });
        for (int i = 0; i < 8; ++i)
            if (!orth[i].empty()) node->children[i] = fut[i].get();
    } else {
        for (int i = 0; i < 8; ++i)
            if (!orth[i].empty())
                node->children[i] = build_tree<F>(orth[i], bounds.create_orthant(i), depth + 1, 0, too_deep);
    }
    F mass = F(0);                                    // :174  Σ m in slice order (recent rustc folds float sums from -0.0, not
                                                      // +0.0: the two differ only when every mass is -0.0, which no caller passes;
                                                      // the vector sum of :175-178 is nalgebra's, folded from zeros())
    for (const Body<F>* p : pts) mass += p->mass;
    F s[3] = {F(0), F(0), F(0)};                      // :175-179  Σ (pos*m) in slice order, then / mass
    for (const Body<F>* p : pts)
        for (int c = 0; c < 3; ++c) s[c] += p->pos[c] * p->mass;
    node->mass = mass;
    for (int c = 0; c < 3; ++c) node->com[c] = s[c] / mass;
    return node;
}

struct WalkCount {
    uint64_t accepted = 0;  // nodes whose monopole was applied
    uint64_t visited = 0;   // calc_force invocations
};

// calc_force (barnes_hut.rs:185-203).  The nested `.sum()` is kept: each level folds its
// children's results left to right from zero.  A leaf (or empty node) that fails the opening
// test has no children and contributes zero -- including the body's own leaf.
template <class F>
void calc_force(const OrthNode<F>& node, const F pos[3], const Settings<F>& s, F out[3], WalkCount& cnt) {
    cnt.visited++;
    F r[3] = {node.com[0] - pos[0], node.com[1] - pos[1], node.com[2] - pos[2]};
    F r2 = norm_sq3(r);
    if (node.bounds.width * node.bounds.width < s.theta2 * r2) {
        F r_dist = std::sqrt(r2 + s.g_soft * s.g_soft);
        F r_cubed = r_dist * r_dist * r_dist;
        F k = s.g * node.mass / r_cubed;
        out[0] = r[0] * k; out[1] = r[1] * k; out[2] = r[2] * k;
        cnt.accepted++;
    } else {
        F acc[3] = {F(0), F(0), F(0)};
        for (int i = 0; i < 8; ++i) {
            if (!node.children[i]) continue;
            F f[3];
            calc_force(*node.children[i], pos, s, f, cnt);
            acc[0] += f[0]; acc[1] += f[1]; acc[2] += f[2];
        }
        out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2];
    }
}

// The OTHER leaf semantics found in the reference ("leaf-mode decision"): the walk
// of src/llm/barnes_hut.rs:915-997 (calc_force_3d) applied to the src/manual tree.  Explicit stack
// there, children pushed in reverse so they pop in orthant order = this recursion's order; ONE running
// sum (force_x += ...); a node closer than r2 < 1e-10 is skipped whole (:933-935; that is how a body
// skips its own leaf); an accepted cell adds g*mass * (1/sqrt(r2+eps2))^3 (:938-954); a leaf that fails
// the opening test is evaluated directly with the same formula unless it is the body itself (:958-972).
template <class F>
void calc_force_direct(const OrthNode<F>& node, const Body<F>* self, const F pos[3], const Settings<F>& s, F acc[3],
                       WalkCount& cnt) {
    cnt.visited++;
    const F dx = node.com[0] - pos[0], dy = node.com[1] - pos[1], dz = node.com[2] - pos[2];
    const F r2 = dx * dx + dy * dy + dz * dz;
    if (r2 < F(1e-10)) return;
    const F g_soft2 = s.g_soft * s.g_soft;
    bool is_leaf = true;
    for (int i = 0; i < 8; ++i) if (node.children[i]) is_leaf = false;
    const bool accept = node.bounds.width * node.bounds.width < s.theta2 * r2;
    if (accept || is_leaf) {
        if (!accept && (node.leaf_body == self || !node.leaf_body)) return;   // own leaf / empty node
        const F r_soft2 = r2 + g_soft2;
        const F inv_r = F(1) / std::sqrt(r_soft2);
        const F inv_r3 = inv_r * inv_r * inv_r;
        const F fm = s.g * node.mass * inv_r3;
        acc[0] += dx * fm; acc[1] += dy * fm; acc[2] += dz * fm;
        cnt.accepted++;
        return;
    }
    for (int i = 0; i < 8; ++i)
        if (node.children[i]) calc_force_direct(*node.children[i], self, pos, s, acc, cnt);
}

template <class F>
std::unique_ptr<OrthNode<F>> build_root(const Body<F>* p, size_t n, const Box3<F>& box, int threads, bool* too_deep) {
    std::vector<const Body<F>*> refs(n);
    for (size_t k = 0; k < n; ++k) refs[k] = &p[k];
    return build_tree<F>(refs, box, 0, threads > 1 ? 2 : 0, too_deep);
}

template <class F>
int bh_update_forces(Body<F>* p, size_t n, const Settings<F>& s, const Box3<F>& box, int threads,
                     uint64_t* accepted, uint64_t* visited, int leaf_mode = 0) {  // barnes_hut.rs:250-263
    bool too_deep = false;
    auto root = build_root(p, n, box, threads, &too_deep);
    if (too_deep) return -2;
    std::vector<WalkCount> counts(std::max(1, threads));
    auto work = [&](int t, size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k) {
            F f[3] = {F(0), F(0), F(0)};
            if (leaf_mode == 1) calc_force_direct(*root, &p[k], p[k].pos, s, f, counts[t]);
            else calc_force(*root, p[k].pos, s, f, counts[t]);
            p[k].acc[0] = f[0]; p[k].acc[1] = f[1]; p[k].acc[2] = f[2];   // overwrite, :260
        }
    };
    if (threads <= 1 || n < 256) work(0, 0, n);
    else {  // par_iter_mut over bodies, :258
        std::vector<std::thread> pool;
        size_t chunk = (n + threads - 1) / threads;
        for (int t = 0; t < threads; ++t) {
            size_t k0 = std::min(n, t * chunk), k1 = std::min(n, k0 + chunk);
            if (k0 < k1) pool.emplace_back(work, t, k0, k1);
        }
        for (auto& th : pool) th.join();
    }
    uint64_t a = 0, v = 0;
    for (auto& c : counts) { a += c.accepted; v += c.visited; }
    if (accepted) *accepted = a;
    if (visited) *visited = v;
    return 0;
}

// Linearise in depth-first pre-order with children in orthant order 0..7 (the order calc_force
// visits them).  skip[i] = index of the first node after i's subtree.
template <class F>
void linearise(const OrthNode<F>& node, const Body<F>* base, std::vector<F>& com_mass, std::vector<F>& width,
               std::vector<int32_t>& skip, std::vector<int32_t>& nchild, std::vector<int32_t>& leaf_body) {
    size_t me = width.size();
    com_mass.push_back(node.com[0]); com_mass.push_back(node.com[1]); com_mass.push_back(node.com[2]);
    com_mass.push_back(node.mass);
    width.push_back(node.bounds.width);
    skip.push_back(0);
    int nc = 0;
    for (auto& c : node.children) if (c) nc++;
    nchild.push_back(nc);
    leaf_body.push_back(node.leaf_body ? int32_t(node.leaf_body - base) : -1);
    for (auto& c : node.children)
        if (c) linearise(*c, base, com_mass, width, skip, nchild, leaf_body);
    skip[me] = int32_t(width.size());
}

// The cells as the reference's Barnes-Hut renderer walks them (barnes_hut.rs:322-343: node.bounds.min() / .max() of every
// node), in the pre-order of linearise: 6 values per node.
template <class F>
void linearise_cells(const OrthNode<F>& node, std::vector<F>& min_max) {
    for (int i = 0; i < 3; ++i) min_max.push_back(node.bounds.center[i] + (-node.bounds.half_width));   // Bounds::min, shared.rs:223-225
    for (int i = 0; i < 3; ++i) min_max.push_back(node.bounds.center[i] + node.bounds.half_width);      // Bounds::max, shared.rs:227-229
    for (auto& c : node.children)
        if (c) linearise_cells(*c, min_max);
}

// f64 diagnostics (not in the reference): KE = Σ ½ m v², PE = −g Σ_{i<j} m_i m_j / sqrt(r²+ε²),
// the potential whose gradient is the force law of brute_force.rs:72-79.
template <class F>
void energy(const Body<F>* p, size_t n, double g, double g_soft, int threads, double* ke, double* pe) {
    double k = 0;
    for (size_t i = 0; i < n; ++i) {
        double v2 = double(p[i].vel[0]) * p[i].vel[0] + double(p[i].vel[1]) * p[i].vel[1] + double(p[i].vel[2]) * p[i].vel[2];
        k += 0.5 * double(p[i].mass) * v2;
    }
    int T = std::max(1, threads);
    std::vector<double> part(T, 0.0);
    auto work = [&](int t) {
        double u = 0;
        for (size_t i = t; i < n; i += T) {
            double ui = 0;
            for (size_t j = 0; j < i; ++j) {
                double dx = double(p[i].pos[0]) - p[j].pos[0], dy = double(p[i].pos[1]) - p[j].pos[1], dz = double(p[i].pos[2]) - p[j].pos[2];
                ui += double(p[j].mass) / std::sqrt(dx * dx + dy * dy + dz * dz + g_soft * g_soft);
            }
            u += ui * double(p[i].mass);
        }
        part[t] = u;
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    double u = 0;
    for (double x : part) u += x;
    *ke = k;
    *pe = -g * u;
}

template <class F>
size_t bh_step_by(Body<F>* p, size_t n, const Settings<F>& s, const Box3<F>& box, F dt, int threads,
                  uint64_t* accepted, uint64_t* visited, int* rc, int leaf_mode = 0) {  // barnes_hut.rs:265-271
    pre_force(p, n, dt);
    n = retain_in_bounds(p, n, box);
    *rc = bh_update_forces(p, n, s, box, threads, accepted, visited, leaf_mode);
    after_force(p, n, dt);
    return n;
}

}  // namespace

// ------------------------------------------------------------------------------ C entry points
// settings arrays are {g, g_soft, dt, theta2}; `aos` is PointParticle<F,3>[n]; box = center[3], width.
#define ORACLE_API(F, SFX)                                                                                     \
    extern "C" void oracle_default_settings_##SFX(F out[4]) {                                                  \
        out[0] = F(1.0); out[1] = F(0.0); out[2] = F(1e-3); out[3] = F(0.5); /* shared.rs:69-78 */             \
    }                                                                                                          \
    extern "C" void oracle_pre_force_##SFX(F* aos, size_t n, F dt) { pre_force((Body<F>*)aos, n, dt); }        \
    extern "C" void oracle_after_force_##SFX(F* aos, size_t n, F dt) { after_force((Body<F>*)aos, n, dt); }    \
    extern "C" size_t oracle_retain_##SFX(F* aos, size_t n, const F c[3], F w) {                               \
        return retain_in_bounds((Body<F>*)aos, n, Box3<F>::make(c, w));                                        \
    }                                                                                                          \
    extern "C" void oracle_bf_update_forces_##SFX(F* aos, size_t n, const F s[4]) {                            \
        bf_update_forces((Body<F>*)aos, n, Settings<F>{s[0], s[1], s[2], s[3]});                               \
    }                                                                                                          \
    extern "C" void oracle_bf_update_forces_rows_##SFX(F* aos, size_t n, const F s[4], int threads) {          \
        bf_update_forces_rows((Body<F>*)aos, n, Settings<F>{s[0], s[1], s[2], s[3]}, threads);                 \
    }                                                                                                          \
    /* rows [row0,row1) only: accelerations of a sample of bodies against all n (full-size checks) */          \
    extern "C" void oracle_bf_update_forces_range_##SFX(F* aos, size_t n, const F s[4], int threads,           \
                                                        size_t row0, size_t row1) {                            \
        bf_update_forces_rows((Body<F>*)aos, n, Settings<F>{s[0], s[1], s[2], s[3]}, threads, row0, row1);     \
    }                                                                                                          \
    extern "C" size_t oracle_bf_step_by_##SFX(F* aos, size_t n, const F s[4], const F c[3], F w, F dt) {       \
        return bf_step_by((Body<F>*)aos, n, Settings<F>{s[0], s[1], s[2], s[3]}, Box3<F>::make(c, w), dt);     \
    }                                                                                                          \
    extern "C" int oracle_bh_update_forces_##SFX(F* aos, size_t n, const F s[4], const F c[3], F w,            \
                                                 int threads, uint64_t* accepted, uint64_t* visited) {         \
        return bh_update_forces((Body<F>*)aos, n, Settings<F>{s[0], s[1], s[2], s[3]}, Box3<F>::make(c, w),    \
                                threads, accepted, visited);                                                   \
    }                                                                                                          \
    /* leaf_mode 1: the src/llm walk (calc_force_direct) on the same tree */                                    \
    extern "C" int oracle_bh_update_forces_mode_##SFX(F* aos, size_t n, const F s[4], const F c[3], F w,       \
                                                      int threads, uint64_t* accepted, uint64_t* visited,      \
                                                      int leaf_mode) {                                         \
        return bh_update_forces((Body<F>*)aos, n, Settings<F>{s[0], s[1], s[2], s[3]}, Box3<F>::make(c, w),    \
                                threads, accepted, visited, leaf_mode);                                        \
    }                                                                                                          \
    extern "C" size_t oracle_bh_step_by_mode_##SFX(F* aos, size_t n, const F s[4], const F c[3], F w, F dt,    \
                                                   int threads, uint64_t* accepted, uint64_t* visited,         \
                                                   int* rc, int leaf_mode) {                                   \
        int r = 0;                                                                                             \
        size_t m = bh_step_by((Body<F>*)aos, n, Settings<F>{s[0], s[1], s[2], s[3]}, Box3<F>::make(c, w), dt,  \
                              threads, accepted, visited, &r, leaf_mode);                                      \
        if (rc) *rc = r;                                                                                       \
        return m;                                                                                              \
    }                                                                                                          \
    extern "C" size_t oracle_bh_step_by_##SFX(F* aos, size_t n, const F s[4], const F c[3], F w, F dt,         \
                                              int threads, uint64_t* accepted, uint64_t* visited, int* rc) {   \
        int r = 0;                                                                                             \
        size_t m = bh_step_by((Body<F>*)aos, n, Settings<F>{s[0], s[1], s[2], s[3]}, Box3<F>::make(c, w), dt,  \
                              threads, accepted, visited, &r);                                                 \
        if (rc) *rc = r;                                                                                       \
        return m;                                                                                              \
    }                                                                                                          \
    /* Returns the node count (or -2 if the depth guard tripped).  Arrays may be null to count only;           \
       otherwise they must hold `cap` nodes (com_mass: 4 scalars per node).  */                                \
    extern "C" long oracle_bh_build_tree_##SFX(const F* aos, size_t n, const F c[3], F w, F* com_mass,         \
                                               F* width, int32_t* skip, int32_t* nchild, int32_t* leaf_body,   \
                                               size_t cap) {                                                   \
        bool too_deep = false;                                                                                 \
        auto root = build_root((const Body<F>*)aos, n, Box3<F>::make(c, w), 1, &too_deep);                     \
        if (too_deep) return -2;                                                                               \
        std::vector<F> cm, wd;                                                                                 \
        std::vector<int32_t> sk, nc, lb;                                                                       \
        linearise(*root, (const Body<F>*)aos, cm, wd, sk, nc, lb);                                             \
        size_t m = wd.size();                                                                                  \
        if (com_mass && m <= cap) {                                                                            \
            std::memcpy(com_mass, cm.data(), cm.size() * sizeof(F));                                           \
            std::memcpy(width, wd.data(), m * sizeof(F));                                                      \
            std::memcpy(skip, sk.data(), m * sizeof(int32_t));                                                 \
            std::memcpy(nchild, nc.data(), m * sizeof(int32_t));                                               \
            std::memcpy(leaf_body, lb.data(), m * sizeof(int32_t));                                            \
        }                                                                                                      \
        return long(m);                                                                                        \
    }                                                                                                          \
    extern "C" long oracle_bh_tree_cells_##SFX(const F* aos, size_t n, const F c[3], F w, F* min_max, size_t cap) {\
        bool too_deep = false;                                                                                 \
        auto root = build_root((const Body<F>*)aos, n, Box3<F>::make(c, w), 1, &too_deep);                     \
        if (too_deep) return -2;                                                                               \
        std::vector<F> mm;                                                                                     \
        linearise_cells(*root, mm);                                                                            \
        if (min_max && mm.size() / 6 <= cap) std::memcpy(min_max, mm.data(), mm.size() * sizeof(F));           \
        return long(mm.size() / 6);                                                                            \
    }                                                                                                          \
    extern "C" void oracle_energy_##SFX(const F* aos, size_t n, double g, double g_soft, int threads,          \
                                        double* ke, double* pe) {                                              \
        energy((const Body<F>*)aos, n, g, g_soft, threads, ke, pe);                                            \
    }

ORACLE_API(float, f32)
ORACLE_API(double, f64)

extern "C" int oracle_hardware_threads() { return int(std::thread::hardware_concurrency()); }
