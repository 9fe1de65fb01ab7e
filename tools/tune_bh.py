"""A/B of the Barnes-Hut walk kernels (per-lane vs wave-cooperative) and the step breakdown."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import knob  # noqa: E402
nb = graft.load_package(tuning=True)   # (the build with the experimental walks and the in-kernel stamps)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ics = nb.plummer(n)
var = knob(nb, "bh_walk_variant")
split = knob(nb, "bh_walk_split")
for math in (nb.FAST,):
    sim = nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BARNES_HUT, math_mode=math)
    sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.25)
    for v, k in ((0, 4), (0, 8), (0, 12), (0, 16), (0, 24), (2, 4), (2, 8), (1, 8), (0, 8)):
        var.value = v; split.value = k
        sim.steps(3); sim.sync()
        sim.set_profiling(True); sim.reset_stats()
        t0 = time.perf_counter()
        sim.steps(20); sim.sync()
        dt = (time.perf_counter() - t0) / 20
        s = sim.stats()
        print(f"math {'fast' if math else 'strict'} walk variant {v} split {k}: step {dt*1e3:.3f} ms; walk kernel {s.force_kernel_ms/s.force_launches:.3f} ms; "
              f"build {s.tree_build_ms/20:.3f} ms; copy+wait {s.tree_copy_ms/20:.3f} ms; visits/step {s.node_visits/20:.3e}; accepted/step {s.interactions/20:.3e}")
    sim.close()
