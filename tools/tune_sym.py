"""A/B of the symmetric all-pairs kernel's occupancy knob and of the one-sided variants, one process."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import knob  # noqa: E402
nb = graft.load_package(tuning=True)   # (the build with the experimental walks and the in-kernel stamps)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ics = nb.plummer(n)
var = knob(nb, "bf_fast_variant")
wpb = knob(nb, "sym_wpb")
rounds = knob(nb, "sym_rounds")
dbg = knob(nb, "sym_debug")
pk = knob(nb, "sym_packed")
sim = nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BRUTE_FORCE, math_mode=nb.FAST)
sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.5)
cases = [("sym", 0, w, r, p) for p in (0, 1) for w, r in ((16, 1), (16, 2), (12, 1), (12, 2), (8, 2), (8, 3))]
res = {c: [] for c in cases}
for rnd in range(3):
    for c in cases:
        var.value, wpb.value, rounds.value, dbg.value, pk.value = c[1], c[2], c[3], 0, c[4]
        sim.update_forces(); sim.sync()
        sim.set_profiling(True); sim.reset_stats()
        import time
        t0 = time.perf_counter()
        for _ in range(20):
            sim.update_forces()
        s = sim.stats()   # drains the stream
        wall = (time.perf_counter() - t0) / 20 * 1e3
        res[c].append((s.force_kernel_ms / s.force_launches, wall))
for c in cases:
    r = sorted(res[c], key=lambda x: x[1])
    print(f"{c}: dominant kernel {min(x[0] for x in r):.4f} ms; whole force pass (3 kernels, wall) min {r[0][1]:.4f} median {r[len(r)//2][1]:.4f} ms -> {n*(n-1)/r[0][1]/1e9:.2f} T interactions/s")

import numpy as np
accs = []
for p in (0, 1):
    var.value, wpb.value, rounds.value, dbg.value, pk.value = 0, 12, 1, 0, p
    sim.update_forces(); sim.sync()
    accs.append(np.array(sim.get_points()["acceleration"], dtype=np.float64))
scale = np.abs(accs[0]).max()
print(f"packed vs unpacked: max |diff| / max |acc| = {np.abs(accs[0] - accs[1]).max() / scale:.3e}")
