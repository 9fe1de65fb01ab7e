"""Mid sizes of the symmetric all-pairs kernel (8 192 .. 49 152 bodies: also every rank's shard at 8 GPUs): kernel time over
(waves per set K, waves per workgroup) -- what make_sym_plan's rule should pick.
    python tools/tune_sym_mid.py [n1,n2,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
nb = graft.load_package()
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8192, 12288, 16384, 24576, 32768, 49152]
for n in sizes:
    ics = nb.plummer(n)
    A = (n + 511) // 512
    L = 8 * ((A + 1) // 2 - 1)
    best = None
    rows = []
    for wpb in (12, 8, 4):
        ks = sorted({k for k in list(range(4, 41, 4)) + [L // d for d in (1, 2, 3, 4, 5, 6, 8)] + [0] if 0 <= k <= max(1, L)})
        for K in ks:
            with nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BRUTE_FORCE, math_mode=nb.FAST, tuning=dict(sym_wpb=wpb, sym_k=K)) as sim:
                sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.25)
                sim.steps(30); sim.sync()
                sim.set_profiling(True); sim.reset_stats()
                t0 = time.perf_counter()
                sim.steps(100); sim.sync()
                step = (time.perf_counter() - t0) / 100
                s = sim.stats()
                ker = s.force_kernel_ms / max(1, s.force_launches)
            rows.append((step, ker, wpb, K))
    rows.sort()
    auto = [r for r in rows if r[3] == 0 and r[2] == 12][0]
    print(f"n={n} A={A} L={L}: rule (wpb 12, K auto): step {auto[0]*1e6:.1f} us kernel {auto[1]*1e3:.1f} us {n*(n-1)/auto[0]:.3e}/s")
    for step, ker, wpb, K in rows[:6]:
        print(f"    wpb {wpb:2d} K {K:3d}: step {step*1e6:7.1f} us  kernel {ker*1e3:7.1f} us  {n*(n-1)/step:.3e} interactions/s", flush=True)
