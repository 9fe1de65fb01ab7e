"""Force-pass time of ONE shard of a world of G (rank 0; the other segments keep their uploaded positions): what each
GPU of a G-GPU run computes per step, on one device.  Sweeps the knobs of the cross-shard plan.

    python tools/tune_sharded.py [n] [--g 8] [--ipt 0,4,8] [--slots 2048,3072,4096] [--wpb 4,12] [--steps 40]
"""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import knob  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("n", nargs="?", type=int, default=65536)
ap.add_argument("--g", default="1,2,4,8")
ap.add_argument("--ipt", default="0")
ap.add_argument("--slots", default="3072")
ap.add_argument("--wpb", default="4")
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--sym-wpb", default="4", help="own-shard kernel: waves per workgroup (4, 8, 12, 16)")
ap.add_argument("--sym-rounds", default="1")
ap.add_argument("--reduce-split", type=int, default=1, help="0: the one-thread-per-body plane reduction of round 1")
a = ap.parse_args()
nb = graft.load_package()
cross = {k: knob(nb, f"cross_{k}") for k in ("ipt", "slots", "wpb")}
sym_wpb, sym_rounds = knob(nb, "sym_wpb"), knob(nb, "sym_rounds")
knob(nb, "sym_reduce_split").value = a.reduce_split
ics = nb.plummer(a.n)
for sw, sr in [(int(x), int(y)) for x in a.sym_wpb.split(",") for y in a.sym_rounds.split(",")]:
  sym_wpb.value, sym_rounds.value = sw, sr
  for G in [int(x) for x in a.g.split(",")]:
      for ipt in [int(x) for x in a.ipt.split(",")]:
          for slots in [int(x) for x in a.slots.split(",")]:
              for wpb in [int(x) for x in a.wpb.split(",")]:
                  cross["ipt"].value, cross["slots"].value, cross["wpb"].value = ipt, slots, wpb
                  sim = nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BRUTE_FORCE, math_mode=nb.FAST, rank=0, world_size=G, capacity=a.n)
                  sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.5)
                  for _ in range(3):
                      nb.sharded_step([sim])
                  sim.sync(); sim.set_profiling(True); sim.reset_stats()
                  t0 = time.perf_counter()
                  for _ in range(a.steps):
                      nb.sharded_step([sim])
                  sim.sync()
                  wall = (time.perf_counter() - t0) / a.steps * 1e3
                  s = sim.stats()
                  k_ms = s.force_kernel_ms / max(1, s.force_launches)
                  print(f"sym_wpb={sw} rounds={sr} G={G} n_own={a.n // G} ipt={ipt} slots={slots} wpb={wpb}: dominant kernel {k_ms:.4f} ms "
                        f"({s.force_kernel_interactions / max(1, s.force_launches) / max(k_ms, 1e-9) / 1e9:.2f} T inter/s), step wall {wall:.4f} ms", flush=True)
                  sim.close()

