"""A/B of the plain fast walk (variant 0) against the LDS-staged walk (variant 3) at the benchmark configuration:
table size, workgroup size and node-range split.  Device tree, so the step is not host-bound.
    python tools/tune_bh_lds.py [n]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import knob  # noqa: E402
nb = graft.load_package(tuning=True)   # (the build with the experimental walks and the in-kernel stamps)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ics = nb.plummer(n)
g = lambda name: knob(nb, name)
var, split, cap, blk = g("nbody_bh_walk_variant"), g("nbody_bh_walk_split"), g("nbody_bh_hot_cap"), g("nbody_bh_walk_lds_block")
sim = nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE)
sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.25)
cases = [(0, 8, 0, 0), (0, 4, 0, 0), (0, 16, 0, 0)]
for k in (8, 4, 16, 2):
    for m in (1024, 2048, 2560, 4096, 4992):
        for b in (1024, 512, 256):
            if m * 32 * (1024 // b) > 160 * 1024 and b != 1024:
                pass
            cases.append((3, k, m, b))
cases.append((0, 8, 0, 0))
for v, k, m, b in cases:
    var.value, split.value = v, k
    if m: cap.value = m
    if b: blk.value = b
    sim.steps(12); sim.sync()          # lets the threshold control settle on this table size
    sim.set_profiling(True); sim.reset_stats()
    t0 = time.perf_counter()
    sim.steps(20); sim.sync()
    dt = (time.perf_counter() - t0) / 20
    s = sim.stats()
    print(f"variant {v} split {k:2d} table {m:5d} block {b:4d}: step {dt*1e3:.3f} ms; walk (incl. prep + reduce) {s.force_kernel_ms/s.force_launches:.3f} ms; "
          f"visits/step {s.node_visits/20:.4e}", flush=True)
sim.close()
