// simulation.hpp -- C++ mirror of the reference's `Simulation` trait (src/shared.rs:80-97) over
// the C ABI of include/nbody_hip.h.  The reference's host language is Rust; this image has no Rust
// toolchain, so the host side above the ABI is restated in C++ with the trait's method names,
// argument meaning and ownership rules (the Rust shim itself is nbody-llm_amd/rust/, untested).
//
//   nbody::PointParticleT<F>      = shared.rs:151-158 (#[repr(C)]; F = float: 40 bytes, F = double: 80 bytes --
//                                   the reference's trait is generic over Float and its driver runs f64)
//   nbody::SimulationSettings     = shared.rs:61-78
//   nbody::Bounds                 = shared.rs:216-243 (center, width)
//   nbody::Simulation             = shared.rs:80-97   (abstract)
//   nbody::IntegratorT            = shared.rs:99-104  (the trait's `I`; LeapFrogIntegratorT = shared.rs:106-149)
//   nbody::BruteForceSimulation   = manual/brute_force.rs:11-103
//   nbody::BarnesHutSimulation    = manual/barnes_hut.rs:93-285
//
// Differences from the Rust trait, all forced by the device boundary:
//   - get_points() returns a const reference to a host vector that is refreshed lazily (one D2H
//     copy when the device state is newer), the same trick the Rust shim plays with UnsafeCell;
//   - settings_mut() hands out a reference and the new values are pushed before the next step;
//   - errors: the reference's signatures are infallible (it panics); here a failed ABI call throws
//     nbody::Error carrying the ABI's code and message.
#pragma once
#include <array>
#include <cstddef>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/nbody_hip.h"

namespace nbody {

template <class F>
struct PointParticleT {  // PointParticle<F, 3>
    F position[3];
    F velocity[3];
    F acceleration[3];
    F mass;
};
using PointParticle = PointParticleT<float>;
using PointParticle64 = PointParticleT<double>;
static_assert(sizeof(PointParticle) == 40 && sizeof(PointParticle64) == 80, "PointParticle must match the reference's #[repr(C)] layout");

template <class F>
struct SimulationSettingsT {  // defaults: shared.rs:69-78
    F g = F(1.0);
    F g_soft = F(0.0);
    F dt = F(1e-3);
    F theta2 = F(0.5);
};
using SimulationSettings = SimulationSettingsT<float>;

template <class F>
struct BoundsT {  // Bounds::new(center, width)
    std::array<F, 3> center{F(0), F(0), F(0)};
    F width = F(1);
};
using Bounds = BoundsT<float>;

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

namespace abi {  // the f32 / f64 forms of the entry points that carry scalars
inline int set_settings(NbodyHandle* h, float g, float e, float dt, float t2) { return nbody_set_settings(h, g, e, dt, t2); }
inline int set_settings(NbodyHandle* h, double g, double e, double dt, double t2) { return nbody_set_settings_f64(h, g, e, dt, t2); }
inline int set_bounds(NbodyHandle* h, const float* c, float w) { return nbody_set_bounds(h, c, w); }
inline int set_bounds(NbodyHandle* h, const double* c, double w) { return nbody_set_bounds_f64(h, c, w); }
inline int step_by(NbodyHandle* h, float dt) { return nbody_step_by(h, dt); }
inline int step_by(NbodyHandle* h, double dt) { return nbody_step_by_f64(h, dt); }
inline int elapsed(const NbodyHandle* h, float* t) { return nbody_elapsed(h, t); }
inline int elapsed(const NbodyHandle* h, double* t) { return nbody_elapsed_f64(h, t); }
template <class F> constexpr int dtype_of() { return sizeof(F) == 8 ? NBODY_F64 : NBODY_F32; }
}  // namespace abi

// The reference's `Integrator<F, D, P>` (shared.rs:99-104): the trait is generic over it.  The device integrates with the
// reference's LeapFrogIntegrator itself (kernels K1 / K3: the only integrator the reference ships, and the fast path).
// Any other integrator runs on the host through SimulationT::step_by_with: the unfused form of step_by -- the device does
// update_forces, the host does integrate_pre_force / retain / integrate_after_force on the synced vector.
template <class F>
struct IntegratorT {
    virtual ~IntegratorT() = default;
    virtual void init() {}
    virtual void integrate_pre_force(std::vector<PointParticleT<F>>& points, F dt) = 0;
    virtual void integrate_after_force(std::vector<PointParticleT<F>>& points, F dt) = 0;
};

// shared.rs:106-149 restated on the host (the same products in the same order: compile without FMA contraction and
// a strict-math run through step_by_with equals the device-integrated run bit for bit)
template <class F>
struct LeapFrogIntegratorT : IntegratorT<F> {
    void integrate_pre_force(std::vector<PointParticleT<F>>& points, F dt) override {
        for (auto& p : points)
            for (int k = 0; k < 3; ++k) p.position[k] += (p.velocity[k] * F(0.5)) * dt;          // :138
    }
    void integrate_after_force(std::vector<PointParticleT<F>>& points, F dt) override {
        for (auto& p : points)
            for (int k = 0; k < 3; ++k) {
                p.velocity[k] += p.acceleration[k] * dt;                                          // :144
                p.position[k] += (p.velocity[k] * F(0.5)) * dt;                                   // :146
            }
    }
};

template <class F>
class SimulationT {
public:
    using Particle = PointParticleT<F>;
    virtual ~SimulationT() { if (h_) nbody_destroy(h_); }
    SimulationT(const SimulationT&) = delete;
    SimulationT& operator=(const SimulationT&) = delete;

    void init() { check(nbody_init(h_)); }                                  // shared.rs:85
    void step() { push_settings(); check(nbody_steps(h_, 1)); dirty_ = true; }   // :86-88
    void steps(int k) { push_settings(); check(nbody_steps(h_, k)); dirty_ = true; }  // k x step(), no host sync
    void step_by(F dt) { push_settings(); check(abi::step_by(h_, dt)); dirty_ = true; }  // :89
    void update_forces() { push_settings(); check(nbody_update_forces(h_)); dirty_ = true; }  // :90
    void add_point(const Particle& p) { check(nbody_add_point(h_, &p)); dirty_ = true; }  // :91
    void remove_point(size_t index) { check(nbody_remove_point(h_, index)); dirty_ = true; }   // :92 (swap_remove)
    const std::vector<Particle>& get_points() const {                       // :93
        if (dirty_) {
            size_t n = 0;
            check(nbody_count(h_, &n));
            points_.resize(n);
            check(nbody_download(h_, points_.data(), n, sizeof(Particle), &n));
            points_.resize(n);
            dirty_ = false;
        }
        return points_;
    }
    // step_by with the integrator `I` of the caller's choice (the trait's generic parameter): brute_force.rs:84-90 /
    // barnes_hut.rs:265-271 step by step -- pre-force on the host, retain (Bounds::contains, shared.rs:210-212: inclusive
    // walls, NaN dropped, order kept), forces on the device, after-force on the host, elapsed += dt
    void step_by_with(IntegratorT<F>& integrator, F dt) {
        push_settings();
        std::vector<Particle> pts = get_points();
        integrator.integrate_pre_force(pts, dt);
        const F hw = bounds_.width * F(0.5);
        size_t kept = 0;
        for (size_t i = 0; i < pts.size(); ++i) {
            bool in = true;
            for (int k = 0; k < 3; ++k) {
                const F lo = bounds_.center[k] + (-hw), hi = bounds_.center[k] + hw;
                if (!(pts[i].position[k] >= lo) || !(pts[i].position[k] <= hi)) in = false;
            }
            if (in) pts[kept++] = pts[i];
        }
        pts.resize(kept);
        check(nbody_upload(h_, pts.data(), pts.size(), sizeof(Particle)));
        check(nbody_update_forces(h_));
        dirty_ = true;
        pts = get_points();
        integrator.integrate_after_force(pts, dt);
        check(nbody_upload(h_, pts.data(), pts.size(), sizeof(Particle)));
        points_ = std::move(pts);
        dirty_ = false;
        host_elapsed_ += dt;
    }
    F elapsed() const { F t = 0; check(abi::elapsed(h_, &t)); return t + host_elapsed_; }   // :94
    const SimulationSettingsT<F>& settings() const { return settings_; }    // :95
    SimulationSettingsT<F>& settings_mut() { settings_dirty_ = true; return settings_; }  // :96
    void sync() { check(nbody_sync(h_)); }
    NbodyStats stats() { NbodyStats s{}; check(nbody_stats(h_, &s)); return s; }
    NbodyHandle* handle() { return h_; }

protected:
    SimulationT(int method, const std::vector<Particle>& points, const BoundsT<F>& bounds, int math_mode,
                size_t capacity, int host_threads, int tree_build = NBODY_TREE_AUTO,
                int leaf_mode = NBODY_LEAF_REFERENCE) : bounds_(bounds) {
        NbodyConfig cfg{};
        cfg.struct_size = sizeof(cfg);
        cfg.method = method;
        cfg.math_mode = math_mode;
        cfg.leaf_mode = leaf_mode;
        cfg.device = -1;
        cfg.rank = 0;
        cfg.world_size = 1;
        cfg.host_threads = host_threads;
        cfg.capacity = capacity ? capacity : (points.empty() ? 1 : points.size());
        cfg.tree_build = tree_build;
        cfg.dtype = abi::dtype_of<F>();
        int rc = nbody_create(&cfg, &h_);
        if (rc) throw Error(rc, nbody_last_error(nullptr));
        check(abi::set_bounds(h_, bounds.center.data(), bounds.width));
        check(nbody_upload(h_, points.data(), points.size(), sizeof(Particle)));
    }
    explicit SimulationT(NbodyHandle* cloned, const SimulationT& src)
        : h_(cloned), settings_(src.settings_), bounds_(src.bounds_) {}
    NbodyHandle* clone_handle() const {                                     // `Clone` supertrait
        NbodyHandle* c = nullptr;
        int rc = nbody_clone(h_, &c);
        if (rc) throw Error(rc, nbody_last_error(nullptr));
        return c;
    }
    void check(int rc) const { if (rc) throw Error(rc, nbody_last_error(h_)); }
    void push_settings() {
        if (!settings_dirty_) return;
        check(abi::set_settings(h_, settings_.g, settings_.g_soft, settings_.dt, settings_.theta2));
        settings_dirty_ = false;
    }

    NbodyHandle* h_ = nullptr;
    SimulationSettingsT<F> settings_{};
    BoundsT<F> bounds_{};
    bool settings_dirty_ = true;
    F host_elapsed_ = F(0);   // advanced by step_by_with (the device's clock only sees its own steps)
    mutable bool dirty_ = true;
    mutable std::vector<Particle> points_;
};
using Simulation = SimulationT<float>;

// Simulation::new(points, LeapFrogIntegrator::new(), bounds) for the two solvers of src/manual
template <class F>
class BruteForceSimulationT : public SimulationT<F> {
public:
    BruteForceSimulationT(const std::vector<PointParticleT<F>>& points, const BoundsT<F>& bounds,
                          int math_mode = NBODY_MATH_FAST, size_t capacity = 0)
        : SimulationT<F>(NBODY_BRUTE_FORCE, points, bounds, math_mode, capacity, 0) {}
    std::unique_ptr<BruteForceSimulationT> clone() const {
        return std::unique_ptr<BruteForceSimulationT>(new BruteForceSimulationT(this->clone_handle(), *this));
    }
private:
    BruteForceSimulationT(NbodyHandle* h, const SimulationT<F>& src) : SimulationT<F>(h, src) {}
};
using BruteForceSimulation = BruteForceSimulationT<float>;

template <class F>
class BarnesHutSimulationT : public SimulationT<F> {
public:
    BarnesHutSimulationT(const std::vector<PointParticleT<F>>& points, const BoundsT<F>& bounds,
                         int math_mode = NBODY_MATH_FAST, size_t capacity = 0, int host_threads = 0,
                         int tree_build = NBODY_TREE_AUTO, int leaf_mode = NBODY_LEAF_REFERENCE)
        : SimulationT<F>(NBODY_BARNES_HUT, points, bounds, math_mode, capacity, host_threads, tree_build, leaf_mode) {}
    std::unique_ptr<BarnesHutSimulationT> clone() const {
        return std::unique_ptr<BarnesHutSimulationT>(new BarnesHutSimulationT(this->clone_handle(), *this));
    }
private:
    BarnesHutSimulationT(NbodyHandle* h, const SimulationT<F>& src) : SimulationT<F>(h, src) {}
};
using BarnesHutSimulation = BarnesHutSimulationT<float>;

}  // namespace nbody
