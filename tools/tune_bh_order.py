"""The walk's node-range split: number of segments K and the order a body group's segments are dispatched in
(index order against nearest-first).  Device tree, N = 65 536 by default.
    python tools/tune_bh_order.py [n]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import knob  # noqa: E402
nb = graft.load_package(tuning=True)   # (the build with the experimental walks and the in-kernel stamps)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ics = nb.plummer(n)
g = lambda name: knob(nb, name)
var, split, order = g("nbody_bh_walk_variant"), g("nbody_bh_walk_split"), g("nbody_bh_walk_order")
sim = nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE)
sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.25)
for v, k, o in [(0, k, 1) for k in (8, 16, 24, 32, 24, 24)]:
    if True:
        var.value = v
        split.value, order.value = k, o
        sim.steps(5); sim.sync()
        sim.set_profiling(True); sim.reset_stats()
        t0 = time.perf_counter()
        sim.steps(30); sim.sync()
        dt = (time.perf_counter() - t0) / 30
        s = sim.stats()
        print(f"variant {v} split {k:2d} order {o}: step {dt*1e3:.3f} ms; walk + reduce {s.force_kernel_ms/s.force_launches:.3f} ms", flush=True)
sim.close()
