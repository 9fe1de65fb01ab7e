"""steps/s and interactions/s of the fast paths across body counts (one GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
nb = graft.load_package()
which = sys.argv[1] if len(sys.argv) > 1 else "bf"   # bf | bh | bhdev (Barnes-Hut, device-side tree build)
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 1048576]
for n in sizes:
    ics = nb.plummer(n)
    sim = nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BRUTE_FORCE if which == "bf" else nb.BARNES_HUT, math_mode=nb.FAST,
                        tree_build=nb.TREE_DEVICE if which == "bhdev" else nb.TREE_HOST)
    sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.25)
    k = max(3, min(200, int(2e11 / (n * n)))) if which == "bf" else max(3, min(100, int(4e6 / n)))
    sim.steps(2); sim.sync()
    sim.set_profiling(True); sim.reset_stats()
    t0 = time.perf_counter()
    sim.steps(k); sim.sync()
    dt = (time.perf_counter() - t0) / k
    s = sim.stats()
    extra = "" if which == "bf" else f" build {s.tree_build_ms/k:.3f} ms walk {s.force_kernel_ms/max(1,s.force_launches):.3f} ms nodes {s.tree_nodes}"
    print(f"{which} n={n:8d}: {dt*1e3:9.4f} ms/step {1/dt:10.1f} steps/s {s.interactions/k/dt:.3e} interactions/s kernel {s.force_kernel_ms/max(1,s.force_launches):.4f} ms{extra}", flush=True)
    sim.close()
