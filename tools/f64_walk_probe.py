"""F = f64 fast Barnes-Hut walk: bodies per lane x body count (Plummer, theta = 0.5, device build).  python tools/f64_walk_probe.py [n,n,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
nb = graft.load_package()
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [65536, 262144, 1048576]
for n in sizes:
    ics = nb.plummer(n, f64=True)
    for bpl in (1, 2, 3, 4, 6, -1):
        sim = nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE, tuning={"bh_walk_duo": bpl})
        sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.25)
        k = max(3, min(40, int(2e6 / n)))
        sim.steps(2); sim.sync()
        sim.set_profiling(True); sim.reset_stats()
        t0 = time.perf_counter()
        sim.steps(k); sim.sync()
        dt = (time.perf_counter() - t0) / k
        s = sim.stats()
        print(f"f64 n={n:8d} bodies/lane {bpl:2d}: walk {s.force_kernel_ms / max(1, s.force_launches):8.4f} ms step {dt * 1e3:8.4f} ms visits/step {s.node_visits / k:.4e}", flush=True)
        sim.close()
