// nbody_cli.cpp -- the reference's headless driver (src/main.rs:31-129) over the HIP engine.
// Keeps the argv contract `-t <threads> -n <points>` that perf_benchmark.py:107-112 drives, the
// disc initial conditions (main.rs:52-89), the hard-coded settings dt = 3e-2, g_soft = 0.02,
// theta2 = 1.0 (main.rs:103-105), the 1000-step loop and the two output lines
//     Elapsed: <duration>
//     Performance: <x> steps/second
// (main.rs:124-128).  Extra flags select what the reference needs a source edit for; --dtype f64 runs the
// reference's own precision (PointParticle<f64,3>, main.rs:52-105); --integrator host steps through the trait's generic
// `Integrator` parameter (shared.rs:99-104) with a leapfrog on the host instead of the device's fused one; --dump FILE
// writes the final PointParticle records.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "simulation.hpp"

static void usage() {
    std::fprintf(stderr,
                 "usage: nbody_cli [-t threads] [-n points] [--method bh|bf] [--ic disc|plummer] [--steps K]\n"
                 "                 [--math fast|strict] [--tree auto|host|device] [--leaf reference|direct]\n"
                 "                 [--dtype f32|f64] [--dt x] [--g-soft x] [--theta2 x]\n"
                 "                 [--width w] [--seed s] [--integrator device|host] [--dump file]\n");
}

template <class F>
static int run(const std::string& method, const std::string& ic, const std::string& math, const std::string& tree,
               const std::string& leaf, size_t threads, size_t num_points, size_t steps, double dt, double g_soft, double theta2,
               double width, unsigned long long seed, const std::string& integrator, const std::string& dump) {
    using P = nbody::PointParticleT<F>;
    const bool wide = sizeof(F) == 8;
    std::vector<P> points;
    if (ic == "disc") {
        points.resize(num_points + 1);  // the star + n disc bodies
        if ((wide ? nbody_ic_disc_f64 : nbody_ic_disc)(points.data(), num_points, sizeof(P), seed)) return 1;
    } else {
        points.resize(num_points);
        if ((wide ? nbody_ic_plummer_f64 : nbody_ic_plummer)(points.data(), num_points, sizeof(P), seed)) return 1;
    }
    const int math_mode = math == "strict" ? NBODY_MATH_STRICT : NBODY_MATH_FAST;
    try {
        std::unique_ptr<nbody::SimulationT<F>> sim;
        nbody::BoundsT<F> bounds{{F(0), F(0), F(0)}, F(width)};
        if (method == "bf") sim.reset(new nbody::BruteForceSimulationT<F>(points, bounds, math_mode));
        else sim.reset(new nbody::BarnesHutSimulationT<F>(points, bounds, math_mode, 0, int(threads),
                                                         tree == "device" ? NBODY_TREE_DEVICE : tree == "host" ? NBODY_TREE_HOST : NBODY_TREE_AUTO,
                                                         leaf == "direct" ? NBODY_LEAF_DIRECT : NBODY_LEAF_REFERENCE));
        sim->settings_mut().dt = F(dt);
        sim->settings_mut().g_soft = F(g_soft);
        sim->settings_mut().theta2 = F(theta2);
        std::printf("Running simulation without rendering...\n");  // main.rs:111
        sim->init();
        auto start = std::chrono::steady_clock::now();
        if (integrator == "host") {   // the trait's generic Integrator, on the host (simulation.hpp step_by_with)
            nbody::LeapFrogIntegratorT<F> leapfrog;
            leapfrog.init();
            for (size_t i = 0; i < steps; ++i) sim->step_by_with(leapfrog, F(dt));
        } else {
            for (size_t i = 0; i < steps; ++i) sim->step();
        }
        sim->sync();
        double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        std::printf("Elapsed: %.6fs\n", secs);
        std::printf("Performance: %.2f steps/second\n", double(steps) / secs);
        NbodyStats st = sim->stats();
        std::printf("Bodies left: %zu  interactions/second: %.4e\n", sim->get_points().size(), double(st.interactions) / secs);
        if (!dump.empty()) {
            std::FILE* f = std::fopen(dump.c_str(), "wb");
            if (!f) { std::fprintf(stderr, "cannot write %s\n", dump.c_str()); return 1; }
            std::fwrite(sim->get_points().data(), sizeof(P), sim->get_points().size(), f);
            std::fclose(f);
        }
    } catch (const nbody::Error& e) {
        std::fprintf(stderr, "nbody error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}

int main(int argc, char** argv) {
    size_t threads = 0, num_points = 10000, steps = 1000;  // main.rs:33-38, :116
    std::string method = "bh", ic = "disc", math = "fast", tree = "auto", leaf = "reference", dtype = "f32", integrator = "device", dump;
    double dt = 3e-2, g_soft = 0.02, theta2 = 1.0, width = 10.0;  // main.rs:59,103-105
    unsigned long long seed = 20250523ull;
    bool width_set = false;
    for (int i = 1; i < argc; ++i) {
        auto next = [&]() -> const char* { if (i + 1 >= argc) { usage(); std::exit(2); } return argv[++i]; };
        if (!std::strcmp(argv[i], "-t") || !std::strcmp(argv[i], "--threads")) threads = std::strtoull(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "-n") || !std::strcmp(argv[i], "--num-points")) num_points = std::strtoull(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--method")) method = next();
        else if (!std::strcmp(argv[i], "--ic")) ic = next();
        else if (!std::strcmp(argv[i], "--math")) math = next();
        else if (!std::strcmp(argv[i], "--tree")) tree = next();
        else if (!std::strcmp(argv[i], "--leaf")) leaf = next();
        else if (!std::strcmp(argv[i], "--dtype")) dtype = next();
        else if (!std::strcmp(argv[i], "--steps")) steps = std::strtoull(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--dt")) dt = std::strtod(next(), nullptr);
        else if (!std::strcmp(argv[i], "--g-soft")) g_soft = std::strtod(next(), nullptr);
        else if (!std::strcmp(argv[i], "--theta2")) theta2 = std::strtod(next(), nullptr);
        else if (!std::strcmp(argv[i], "--width")) { width = std::strtod(next(), nullptr); width_set = true; }
        else if (!std::strcmp(argv[i], "--seed")) seed = std::strtoull(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--integrator")) integrator = next();
        else if (!std::strcmp(argv[i], "--dump")) dump = next();
        else { usage(); return 2; }
    }
    if (ic == "plummer" && !width_set) width = 64.0;
    if (dtype == "f64") return run<double>(method, ic, math, tree, leaf, threads, num_points, steps, dt, g_soft, theta2, width, seed, integrator, dump);
    return run<float>(method, ic, math, tree, leaf, threads, num_points, steps, dt, g_soft, theta2, width, seed, integrator, dump);
}
