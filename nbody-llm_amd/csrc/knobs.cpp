// knobs.cpp -- whose knobs the launchers read: the handle the calling thread is serving (kernels.h, struct Tuning).
#include "kernels.h"

#include <algorithm>
#include <cmath>

namespace nbody {

namespace {
const Tuning kDefaults{};
thread_local const Tuning* t_current = nullptr;
}  // namespace

const Tuning& tuning() { return t_current ? *t_current : kDefaults; }
void bind_tuning(const Tuning* t) { t_current = t; }


WalkPlan walk_plan(size_t n_order, bool fast_math, int max_segments, float theta2) {
    const Tuning& t = tuning();
    const size_t waves = std::max<size_t>(1, (n_order + 63) / 64);   // at one body per lane
    WalkPlan p{1, 1};
    if (fast_math) {
        if (t.bh_walk_duo >= 2) p.bodies_per_lane = t.bh_walk_duo >= 8 ? 8 : t.bh_walk_duo >= 6 ? 6 : t.bh_walk_duo >= 4 ? 4 : t.bh_walk_duo;
        else if (t.bh_walk_duo < 0) {
            // Sharing pays with the length of a body's walk (the far field is what neighbours have in common) and with the
            // number of lane groups left to fill the chip.  The thresholds were measured on Plummer spheres at theta = 0.5
            // (1 830 visits per body at 65 536 bodies); a walk's length goes like theta^-3, so other opening angles count as
            // a body number scaled by that (the reference driver's disc at theta = 1: no sharing at 100 000 bodies, 3 per lane
            // at 10^6 -- both as measured).
            const double scale = std::min(4.0, std::max(1.0 / 64.0, std::pow(0.25 / std::max(1e-6, double(theta2)), 1.5)));
            const double n_eff = double(n_order) * scale;
            p.bodies_per_lane = n_eff < 24576 ? 1 : n_eff < 196608 ? 2 : n_eff < 393216 ? 3 : n_eff < 786432 ? 4 : 6;
        }
    }
    if (t.bh_walk_split > 0) p.segments = t.bh_walk_split;
    else if (p.bodies_per_lane > 1) p.segments = int(std::min<size_t>(64, (size_t(32768) * p.bodies_per_lane + waves / 2) / waves));
    else p.segments = int((16384 + waves - 1) / waves);
    p.segments = std::max(1, std::min(max_segments, p.segments));
    return p;
}
}  // namespace nbody
