// simulation.hpp -- C++ mirror of the reference's `Simulation` trait (src/shared.rs:80-97) over
// the C ABI of include/nbody_hip.h.  The reference's host language is Rust; this image has no Rust
// toolchain, so the host side above the ABI is restated in C++ with the trait's method names,
// argument meaning and ownership rules (the Rust shim itself is nbody-llm_amd/rust/, untested).
//
//   nbody::PointParticle          = shared.rs:151-158 (#[repr(C)], 40 bytes)
//   nbody::SimulationSettings     = shared.rs:61-78
//   nbody::Bounds                 = shared.rs:216-243 (center, width)
//   nbody::Simulation             = shared.rs:80-97   (abstract)
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

struct PointParticle {  // PointParticle<f32, 3>
    float position[3];
    float velocity[3];
    float acceleration[3];
    float mass;
};
static_assert(sizeof(PointParticle) == 40, "PointParticle must match the reference's #[repr(C)] layout");

struct SimulationSettings {  // defaults: shared.rs:69-78
    float g = 1.0f;
    float g_soft = 0.0f;
    float dt = 1e-3f;
    float theta2 = 0.5f;
};

struct Bounds {  // Bounds::new(center, width)
    std::array<float, 3> center{0.f, 0.f, 0.f};
    float width = 1.f;
};

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

class Simulation {
public:
    virtual ~Simulation() { if (h_) nbody_destroy(h_); }
    Simulation(const Simulation&) = delete;
    Simulation& operator=(const Simulation&) = delete;

    void init() { check(nbody_init(h_)); }                                  // shared.rs:85
    void step() { push_settings(); check(nbody_steps(h_, 1)); dirty_ = true; }   // :86-88
    void steps(int k) { push_settings(); check(nbody_steps(h_, k)); dirty_ = true; }  // k x step(), no host sync
    void step_by(float dt) { push_settings(); check(nbody_step_by(h_, dt)); dirty_ = true; }  // :89
    void update_forces() { push_settings(); check(nbody_update_forces(h_)); dirty_ = true; }  // :90
    void add_point(const PointParticle& p) { check(nbody_add_point(h_, &p)); dirty_ = true; }  // :91
    void remove_point(size_t index) { check(nbody_remove_point(h_, index)); dirty_ = true; }   // :92 (swap_remove)
    const std::vector<PointParticle>& get_points() const {                  // :93
        if (dirty_) {
            size_t n = 0;
            check(nbody_count(h_, &n));
            points_.resize(n);
            check(nbody_download(h_, points_.data(), n, sizeof(PointParticle), &n));
            points_.resize(n);
            dirty_ = false;
        }
        return points_;
    }
    float elapsed() const { float t = 0; check(nbody_elapsed(h_, &t)); return t; }   // :94
    const SimulationSettings& settings() const { return settings_; }        // :95
    SimulationSettings& settings_mut() { settings_dirty_ = true; return settings_; }  // :96
    void sync() { check(nbody_sync(h_)); }
    NbodyStats stats() { NbodyStats s{}; check(nbody_stats(h_, &s)); return s; }
    NbodyHandle* handle() { return h_; }

protected:
    Simulation(int method, const std::vector<PointParticle>& points, const Bounds& bounds, int math_mode,
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
        int rc = nbody_create(&cfg, &h_);
        if (rc) throw Error(rc, nbody_last_error(nullptr));
        check(nbody_set_bounds(h_, bounds.center.data(), bounds.width));
        check(nbody_upload(h_, points.data(), points.size(), sizeof(PointParticle)));
    }
    explicit Simulation(NbodyHandle* cloned, const Simulation& src)
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
        check(nbody_set_settings(h_, settings_.g, settings_.g_soft, settings_.dt, settings_.theta2));
        settings_dirty_ = false;
    }

    NbodyHandle* h_ = nullptr;
    SimulationSettings settings_{};
    Bounds bounds_{};
    bool settings_dirty_ = true;
    mutable bool dirty_ = true;
    mutable std::vector<PointParticle> points_;
};

// Simulation::new(points, LeapFrogIntegrator::new(), bounds) for the two solvers of src/manual
class BruteForceSimulation : public Simulation {
public:
    BruteForceSimulation(const std::vector<PointParticle>& points, const Bounds& bounds,
                         int math_mode = NBODY_MATH_FAST, size_t capacity = 0)
        : Simulation(NBODY_BRUTE_FORCE, points, bounds, math_mode, capacity, 0) {}
    std::unique_ptr<BruteForceSimulation> clone() const {
        return std::unique_ptr<BruteForceSimulation>(new BruteForceSimulation(clone_handle(), *this));
    }
private:
    BruteForceSimulation(NbodyHandle* h, const Simulation& src) : Simulation(h, src) {}
};

class BarnesHutSimulation : public Simulation {
public:
    BarnesHutSimulation(const std::vector<PointParticle>& points, const Bounds& bounds,
                        int math_mode = NBODY_MATH_FAST, size_t capacity = 0, int host_threads = 0,
                        int tree_build = NBODY_TREE_AUTO, int leaf_mode = NBODY_LEAF_REFERENCE)
        : Simulation(NBODY_BARNES_HUT, points, bounds, math_mode, capacity, host_threads, tree_build, leaf_mode) {}
    std::unique_ptr<BarnesHutSimulation> clone() const {
        return std::unique_ptr<BarnesHutSimulation>(new BarnesHutSimulation(clone_handle(), *this));
    }
private:
    BarnesHutSimulation(NbodyHandle* h, const Simulation& src) : Simulation(h, src) {}
};

}  // namespace nbody
