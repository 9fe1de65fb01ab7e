// nbody_cli.cpp -- the reference's headless driver (src/main.rs:31-129) over the HIP engine.
// Keeps the argv contract `-t <threads> -n <points>` that perf_benchmark.py:107-112 drives, the
// disc initial conditions (main.rs:52-89), the hard-coded settings dt = 3e-2, g_soft = 0.02,
// theta2 = 1.0 (main.rs:103-105), the 1000-step loop and the two output lines
//     Elapsed: <duration>
//     Performance: <x> steps/second
// (main.rs:124-128).  Extra flags select what the reference needs a source edit for.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "simulation.hpp"

static void usage() {
    std::fprintf(stderr,
                 "usage: nbody_cli [-t threads] [-n points] [--method bh|bf] [--ic disc|plummer] [--steps K]\n"
                 "                 [--math fast|strict] [--tree auto|host|device] [--leaf reference|direct]\n"
                 "                 [--dt x] [--g-soft x] [--theta2 x]\n"
                 "                 [--width w] [--seed s]\n");
}

int main(int argc, char** argv) {
    size_t threads = 0, num_points = 10000, steps = 1000;  // main.rs:33-38, :116
    std::string method = "bh", ic = "disc", math = "fast", tree = "auto", leaf = "reference";
    float dt = 3e-2f, g_soft = 0.02f, theta2 = 1.0f, width = 10.0f;  // main.rs:59,103-105
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
        else if (!std::strcmp(argv[i], "--steps")) steps = std::strtoull(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--dt")) dt = std::strtof(next(), nullptr);
        else if (!std::strcmp(argv[i], "--g-soft")) g_soft = std::strtof(next(), nullptr);
        else if (!std::strcmp(argv[i], "--theta2")) theta2 = std::strtof(next(), nullptr);
        else if (!std::strcmp(argv[i], "--width")) { width = std::strtof(next(), nullptr); width_set = true; }
        else if (!std::strcmp(argv[i], "--seed")) seed = std::strtoull(next(), nullptr, 10);
        else { usage(); return 2; }
    }
    if (ic == "plummer" && !width_set) width = 64.0f;

    std::vector<nbody::PointParticle> points;
    if (ic == "disc") {
        points.resize(num_points + 1);  // the star + n disc bodies
        if (nbody_ic_disc(points.data(), num_points, sizeof(nbody::PointParticle), seed)) return 1;
    } else {
        points.resize(num_points);
        if (nbody_ic_plummer(points.data(), num_points, sizeof(nbody::PointParticle), seed)) return 1;
    }
    const int math_mode = math == "strict" ? NBODY_MATH_STRICT : NBODY_MATH_FAST;
    try {
        std::unique_ptr<nbody::Simulation> sim;
        nbody::Bounds bounds{{0.f, 0.f, 0.f}, width};
        if (method == "bf") sim.reset(new nbody::BruteForceSimulation(points, bounds, math_mode));
        else sim.reset(new nbody::BarnesHutSimulation(points, bounds, math_mode, 0, int(threads),
                                                      tree == "device" ? NBODY_TREE_DEVICE : tree == "host" ? NBODY_TREE_HOST : NBODY_TREE_AUTO,
                                                      leaf == "direct" ? NBODY_LEAF_DIRECT : NBODY_LEAF_REFERENCE));
        sim->settings_mut().dt = dt;
        sim->settings_mut().g_soft = g_soft;
        sim->settings_mut().theta2 = theta2;
        std::printf("Running simulation without rendering...\n");  // main.rs:111
        sim->init();
        auto start = std::chrono::steady_clock::now();
        for (size_t i = 0; i < steps; ++i) sim->step();
        sim->sync();
        double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        std::printf("Elapsed: %.6fs\n", secs);
        std::printf("Performance: %.2f steps/second\n", double(steps) / secs);
        NbodyStats st = sim->stats();
        std::printf("Bodies left: %zu  interactions/second: %.4e\n", sim->get_points().size(), double(st.interactions) / secs);
    } catch (const nbody::Error& e) {
        std::fprintf(stderr, "nbody error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
