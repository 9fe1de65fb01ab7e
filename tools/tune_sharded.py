"""Force-pass time of ONE shard of a world of G (rank 0; the other segments keep their uploaded
positions), one-sided kernel variants.  Emulates what each GPU of a G-GPU run computes per step."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
nb = graft.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
var = ctypes.c_int.in_dll(nb.lib, "nbody_bf_fast_variant")
ics = nb.plummer(n)
for G in (1, 2, 4, 8):
    sim = nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BRUTE_FORCE, math_mode=nb.FAST, rank=0, world_size=G, capacity=n)
    sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.5)
    for v in variants:
        var.value = v
        if G == 1 and v == 0:
            pass
        for _ in range(3):
            nb.sharded_step([sim])
        sim.sync(); sim.set_profiling(True); sim.reset_stats()
        t0 = time.perf_counter()
        for _ in range(20):
            nb.sharded_step([sim])
        s = sim.stats()
        wall = (time.perf_counter() - t0) / 20 * 1e3
        print(f"G={G} n_own={n//G} variant {v}: force kernel {s.force_kernel_ms/s.force_launches:.4f} ms; step wall {wall:.4f} ms "
              f"-> per-GPU {s.force_kernel_interactions/s.force_launches/(s.force_kernel_ms/s.force_launches)/1e9:.2f} T/s in-kernel")
    sim.close()
var.value = 0
