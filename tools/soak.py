import sys, time
sys.path.insert(0, "/root/repo")
import __graft_entry__ as graft
nb = graft.load_package()
import numpy as np
for method, n, steps in ((nb.BRUTE_FORCE, 8192, 20000), (nb.BARNES_HUT, 8192, 6000)):
    for tb in ((nb.TREE_HOST, nb.TREE_DEVICE) if method == nb.BARNES_HUT else (nb.TREE_HOST,)):
        sim = nb.Simulation(nb.plummer(n), (0, 0, 0), 64.0, method=method, math_mode=nb.FAST, tree_build=tb)
        sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.25)
        sim.init(); sim.set_profiling(True)
        e0 = sim.energy()
        t0 = time.perf_counter()
        sim.steps(steps)
        s = sim.stats()
        e1 = sim.energy()
        print(f"method {method} tree {tb}: {steps} steps in {time.perf_counter()-t0:.2f} s, launches timed {s.force_launches}, bodies {len(sim)}, energy drift {(sum(e1)-sum(e0))/abs(sum(e0)):.2e}")
        sim.close()
