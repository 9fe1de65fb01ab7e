"""Barnes-Hut fast walk: bodies per lane (bh_walk_duo) x node-range segments (bh_walk_split) across body counts, one GPU.
Prints the walk kernel's time and the step time.   python tools/tune_bh_duo.py [n,n,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
nb = graft.load_package()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
DISC = "--disc" in sys.argv   # the reference driver's workload (src/main.rs): disc around a star, box 10, theta2 = 1, g_soft 0.02, dt 3e-2
sizes = [int(x) for x in args[0].split(",")] if args else [16384, 32768, 65536, 131072, 262144, 1048576, 4194304]
for n in sizes:
    ics = nb.disc(n) if DISC else nb.plummer(n)
    waves = (n + 63) // 64
    rows = []
    for bpl in (1, 2, 3, 4, 6, 8):
        if bpl > 1 and n < 16384:
            continue
        for target in (4096, 8192, 16384, 32768, 65536):
            K = max(1, min(256, -(-target * bpl // waves)))
            if any(r[0] == bpl and r[1] == K for r in rows):
                continue
            sim = nb.Simulation(ics, (0, 0, 0), 10.0 if DISC else 64.0, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE,
                                tuning={"bh_walk_duo": bpl, "bh_walk_split": K})
            sim.settings = nb.Settings(1.0, 0.02, 3e-2, 1.0) if DISC else nb.Settings(1.0, 1e-2, 1e-3, 0.25)
            k = max(3, min(60, int(3e6 / n)))
            sim.steps(2); sim.sync()
            sim.set_profiling(True); sim.reset_stats()
            t0 = time.perf_counter()
            sim.steps(k); sim.sync()
            dt = (time.perf_counter() - t0) / k
            s = sim.stats()
            rows.append((bpl, K, s.force_kernel_ms / max(1, s.force_launches), dt * 1e3))
            sim.close()
    def run(tuning):
        sim = nb.Simulation(ics, (0, 0, 0), 10.0 if DISC else 64.0, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE, tuning=tuning)
        sim.settings = nb.Settings(1.0, 0.02, 3e-2, 1.0) if DISC else nb.Settings(1.0, 1e-2, 1e-3, 0.25)
        k = max(3, min(60, int(3e6 / n)))
        sim.steps(2); sim.sync()
        sim.set_profiling(True); sim.reset_stats()
        t0 = time.perf_counter()
        sim.steps(k); sim.sync()
        dt = (time.perf_counter() - t0) / k
        s = sim.stats()
        sim.close()
        return s.force_kernel_ms / max(1, s.force_launches), dt * 1e3
    auto = run({})
    best = min(rows, key=lambda r: r[3])
    for r in sorted(rows, key=lambda r: r[3])[:6]:
        print(f"n={n:8d} bodies/lane {r[0]} segments {r[1]:3d}: walk {r[2]:8.4f} ms step {r[3]:8.4f} ms{'  <-- best' if r is best else ''}", flush=True)
    plain = min((r for r in rows if r[0] == 1), key=lambda r: r[3])
    print(f"n={n:8d} best plain: segments {plain[1]} walk {plain[2]:.4f} step {plain[3]:.4f}; gain {plain[3] / best[3]:.3f}x; the library's own plan: walk {auto[0]:.4f} step {auto[1]:.4f}", flush=True)
