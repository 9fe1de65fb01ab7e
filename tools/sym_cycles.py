"""Diagnostic: in-kernel cycle stamps of the symmetric all-pairs kernel (DBG build path 4):
shader cycles per rotation step and the clock the chip holds, per wave."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import knob  # noqa: E402
nb = graft.load_package(tuning=True)   # (the build with the experimental walks and the in-kernel stamps)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wps_list = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [12]
dbg_list = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [4]
ics = nb.plummer(n)
wps = knob(nb, "sym_wpb")   # waves per workgroup (16, 12 or 8)
dbg = knob(nb, "sym_debug")
sim = nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BRUTE_FORCE, math_mode=nb.FAST)
sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.5)
for w, dv in [(w, d) for w in wps_list for d in dbg_list]:
    wps.value, dbg.value = w, dv
    mode = os.environ.get('SYM_CYCLES_MODE', 'forces')
    if mode == 'steps':
        sim.steps(200)
    else:
        for _ in range(200):   # ~0.2 s of back-to-back launches so the clock settles
            sim.update_forces()
    sim.sync()
    A = (n + 511) // 512
    K = min(126, (256 * w) // A)   # one round of CU-sized workgroups (the plan's choice at N = 65 536)
    nw = min(8192, A * K)
    buf = (ctypes.c_ulonglong * (3 * nw))()
    assert nb.lib.nbody_sym_read_stamps(buf, nw) == 0
    raw = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 3)
    st = raw.astype(np.float64)
    st[:, 2] = (raw[:, 2] & np.uint64(0xFFFF)).astype(np.float64)
    steps = st[:, 2] * 64
    cyc_per_step = st[:, 0] / steps
    ghz = st[:, 0] / st[:, 1] * 0.1
    print(f"dbg {dv} waves/workgroup {w}: waves {nw}, chunks/wave {st[:,2].min():.0f}-{st[:,2].max():.0f}, "
          f"cycles per 8-pair step: median {np.median(cyc_per_step):.1f} (min {cyc_per_step.min():.1f}, max {cyc_per_step.max():.1f}); "
          f"clock median {np.median(ghz):.3f} GHz; wave lifetime min {st[:,1].min()/100:.1f} median {np.median(st[:,1])/100:.1f} us max {st[:,1].max()/100:.1f} us")
dbg.value = 0
