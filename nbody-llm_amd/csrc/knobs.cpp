// knobs.cpp -- whose knobs the launchers read: the handle the calling thread is serving (kernels.h, struct Tuning).
#include "kernels.h"

namespace nbody {

namespace {
const Tuning kDefaults{};
thread_local const Tuning* t_current = nullptr;
}  // namespace

const Tuning& tuning() { return t_current ? *t_current : kDefaults; }
void bind_tuning(const Tuning* t) { t_current = t; }

}  // namespace nbody
