"""Times every instantiation of the fast all-pairs kernel at N = 65 536 (one process, interleaved
rounds; cdna_hip_programming.md rule 24)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import knob  # noqa: E402
nb = graft.load_package(tuning=True)   # (the build with the experimental walks and the in-kernel stamps)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4]
ics = nb.plummer(n)
var = knob(nb, "bf_fast_variant")
sim = nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BRUTE_FORCE, math_mode=nb.FAST)
sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.5)
res = {v: [] for v in variants}
for rnd in range(4):
    for v in variants:
        var.value = v
        sim.update_forces(); sim.sync()
        sim.set_profiling(True); sim.reset_stats()
        for _ in range(10):
            sim.update_forces()
        s = sim.stats()
        res[v].append(s.force_kernel_ms / s.force_launches)
for v in variants:
    r = sorted(res[v])
    print(f"variant {v:4d}: min {r[0]:.4f} ms  median {r[len(r)//2]:.4f} ms  -> {n*(n-1)/r[0]/1e9:.1f} G interactions/s")
