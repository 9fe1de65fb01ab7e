"""A/B of the symmetric all-pairs kernel's occupancy knob and of the one-sided variants, one process."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
nb = graft.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ics = nb.plummer(n)
var = ctypes.c_int.in_dll(nb.lib, "nbody_bf_fast_variant")
wps = ctypes.c_int.in_dll(nb.lib, "nbody_sym_waves_per_simd")
dbg = ctypes.c_int.in_dll(nb.lib, "nbody_sym_debug")
sim = nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BRUTE_FORCE, math_mode=nb.FAST)
sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.5)
cases = [("sym", 0, 4, 0), ("sym", 0, 6, 0), ("directed", 4, 4, 0)]
res = {c: [] for c in cases}
for rnd in range(3):
    for c in cases:
        var.value, wps.value, dbg.value = c[1], c[2], c[3]
        sim.update_forces(); sim.sync()
        sim.set_profiling(True); sim.reset_stats()
        for _ in range(10):
            sim.update_forces()
        s = sim.stats()
        res[c].append(s.force_kernel_ms / s.force_launches)
for c in cases:
    r = sorted(res[c])
    print(f"{c}: min {r[0]:.4f} ms median {r[len(r)//2]:.4f} ms -> {n*(n-1)/r[0]/1e9:.2f} T interactions/s")
