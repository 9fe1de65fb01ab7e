"""k_bf_sym: 8 against 4 bodies per lane of a resident set (Tuning::sym_ipt) across body counts.  python tools/sym_ipt_probe.py"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
nb = graft.load_package()
for n in (8192, 10240, 12288, 16384, 20480, 24576, 32768, 49152, 65536):
    ics = nb.plummer(n)
    for ipt in (8, 4):
        sim = nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BRUTE_FORCE, math_mode=nb.FAST, tuning={"sym_ipt": ipt})
        sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.25)
        k = max(50, min(2000, int(4e11 / (n * n))))
        sim.steps(50); sim.sync()
        sim.set_profiling(True); sim.reset_stats()
        t0 = time.perf_counter(); sim.steps(k); sim.sync(); dt = (time.perf_counter() - t0) / k
        s = sim.stats()
        print(f"n={n:6d} ipt={ipt}: step {dt*1e6:8.2f} us kernel {1e3*s.force_kernel_ms/max(1,s.force_launches):8.2f} us {s.interactions/k/dt:.3e} interactions/s", flush=True)
        sim.close()
