"""Diagnostic for the cooperative walk (variant 4): per wave start/end inside one launch, iterations, window fills,
time spent in the entry replay.    python tools/coop_wave_times.py [n] [K,K,...]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import knob  # noqa: E402
nb = graft.load_package(tuning=True)   # (the build with the experimental walks and the in-kernel stamps)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
splits = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [48]
g = lambda name: knob(nb, name)
dbg, split, var = g("nbody_bh_walk_debug"), g("nbody_bh_walk_split"), g("nbody_bh_walk_variant")
var.value = int(os.environ.get('VARIANT', '5'))
ics = nb.plummer(n)
sim = nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE)
sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.25)
groups = (n + 63) // 64
for K in splits:
    split.value = K
    dbg.value = 0
    for _ in range(20):
        sim.update_forces()
    sim.sync()
    dbg.value = 1
    sim.update_forces(); sim.sync()
    dbg.value = 0
    nw = min(65536, K * groups)
    buf = (ctypes.c_ulonglong * (3 * nw))()
    assert nb.lib.nbody_bh_read_stamps(buf, nw) == 0
    raw = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 3)
    it = (raw[:, 2] & np.uint64(0xFFFFFF)).astype(np.float64)
    fills = ((raw[:, 2] >> np.uint64(24)) & np.uint64(0xFFFFF)).astype(np.float64)
    entry = ((raw[:, 2] >> np.uint64(44)) & np.uint64(0xFFFF)).astype(np.float64) / 100.0   # us
    t0 = raw[:, 0].min()
    beg, end = (raw[:, 0] - t0).astype(np.float64) / 100.0, (raw[:, 1] - t0).astype(np.float64) / 100.0
    life = end - beg
    rank = np.arange(nw) // groups      # dispatch rank of the segment (0 = the group's own)
    span = end.max()
    print(f"K={K}: {nw} waves, launch span {span:.1f} us; starts: median {np.median(beg):.1f} 90% {np.quantile(beg, .9):.1f} max {beg.max():.1f}; "
          f"ends: 50% {np.median(end):.1f} 90% {np.quantile(end, .9):.1f} 99% {np.quantile(end, .99):.1f}")
    print(f"   iterations/wave: mean {it.mean():.0f} median {np.median(it):.0f} 99% {np.quantile(it, .99):.0f} max {it.max():.0f}  sum {it.sum():.3e}; "
          f"window fills/wave mean {fills.mean():.1f} (one per {it.sum() / max(1, fills.sum()):.1f} iterations); entry replay mean {entry.mean():.2f} us max {entry.max():.2f} us")
    long = it > np.quantile(it, 0.99)
    cyc = life * 2400.0
    print(f"   longest 1% of waves: lifetime {np.median(life[long]):.1f} us, {np.median(cyc[long] / it[long]):.0f} cycles/iteration; "
          f"all waves: sum of lifetimes {life.sum() / 1e3:.1f} ms = {life.sum() / span / 8192:.2f} of the chip's wave slots over the span")
    for r in (0, 1, 2, 4, 8, 16, K - 1):
        m = rank == r
        if m.any():
            print(f"   segment rank {r:2d}: iterations mean {it[m].mean():6.0f} max {it[m].max():6.0f}; start median {np.median(beg[m]):6.1f} us; lifetime median {np.median(life[m]):6.1f} max {life[m].max():6.1f} us; cycles/iteration {np.median(cyc[m] / np.maximum(it[m], 1)):.0f}")
    alive = [(end > f * span).sum() for f in (0.25, 0.5, 0.75, 0.9)]
    print("   waves not yet finished at 25/50/75/90 % of the span:", alive)
split.value = 0
var.value = 0
