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

# ---- round 2: the paths added since (long runs: no error, no loss of bodies, bounded energy drift, stable memory)
import ctypes


def gpu_mem_used():
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    return (total.value - free.value) / 2 ** 20


# spatial shards: 4 ranks, a box the cluster slowly leaks out of, 600 steps
n, G = 20000, 4
ics = nb.plummer(n, seed=5)
box = ((0.0, 0.0, 0.0), 6.0)
st = nb.Settings(1.0, 1e-2, 2e-3, 0.25)
sims = [nb.Simulation(ics, *box, method=nb.BARNES_HUT, math_mode=nb.FAST, rank=r, world_size=G, capacity=n, shard_mode=nb.SHARD_SPATIAL) for r in range(G)]
for s in sims:
    s.settings = st
    s.init()
t0 = time.perf_counter()
m0 = None
for k in range(600):
    nb.spatial_step(sims)
    if k == 50:
        m0 = gpu_mem_used()
owned = [len(s) for s in sims]
rec, idx = nb.spatial_gather(sims, n)
mig = sum(s.let_stats().bodies_migrated for s in sims)
print(f"spatial shards: 600 steps in {time.perf_counter()-t0:.1f} s, bodies {sum(owned)} of {n} (per rank {owned}), migrated in all {mig}, "
      f"GPU memory {m0:.0f} -> {gpu_mem_used():.0f} MiB, ids unique {len(np.unique(idx)) == len(idx)}")
for s in sims:
    s.close()

# f64 with the device build, 3000 steps
ics64 = nb.plummer(8192, seed=6, f64=True)
with nb.Simulation(ics64, (0, 0, 0), 64.0, method=nb.BARNES_HUT, tree_build=nb.TREE_DEVICE) as sim:
    sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.25)
    sim.init()
    e0 = sim.energy()
    t0 = time.perf_counter()
    sim.steps(3000)
    e1 = sim.energy()
    print(f"f64 device tree: 3000 steps in {time.perf_counter()-t0:.2f} s, bodies {len(sim)}, energy drift {(sum(e1)-sum(e0))/abs(sum(e0)):.2e}")

# sharded brute force, 4 handles, 1500 steps
ics = nb.plummer(16384, seed=7)
sims = [nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BRUTE_FORCE, math_mode=nb.FAST, rank=r, world_size=4, capacity=16384) for r in range(4)]
for s in sims:
    s.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.5)
    s.init()
t0 = time.perf_counter()
for _ in range(1500):
    nb.sharded_step(sims)
got = np.concatenate([s.get_points() for s in sims])
ke = 0.5 * (got["mass"] * (got["velocity"].astype(np.float64) ** 2).sum(1)).sum()
print(f"sharded brute force: 1500 steps in {time.perf_counter()-t0:.1f} s, bodies {len(got)}, kinetic energy {ke:.4f} (virial 0.25 expected)")
for s in sims:
    s.close()
