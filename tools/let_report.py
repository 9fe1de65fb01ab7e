#!/usr/bin/env python3
"""Spatial-shard Barnes-Hut (NBODY_SHARD_SPATIAL) emulated on ONE GPU: G handles play G ranks, the four exchanges are
device-to-device copies.  Prints, per rank and per step: bodies owned, nodes built / exported / imported, bytes sent
against the all-gather of positions, and the device time of each phase (HIP events); beside it the single-GPU device-tree
step of the same bodies (what every rank of the replicate-everything scheme has to do for the build).

    python tools/let_report.py [--n 4194304] [--gpus 8] [--steps 4] [--json profiles/r02_let_emulation.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 22)
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--theta", type=float, default=0.5)
    ap.add_argument("--json", default=None)
    ap.add_argument("--balance", default="work", choices=["work", "count"])
    a = ap.parse_args()
    nb = graft.load_package()
    box = ((0.0, 0.0, 0.0), 64.0)
    st = nb.Settings(1.0, 1e-2, 1e-3, a.theta * a.theta)
    ics = nb.plummer(a.n, seed=13)
    G = a.gpus

    with nb.Simulation(ics, *box, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE) as one:
        one.settings = st
        one.init()
        one.steps(2)
        one.sync()
        one.set_profiling(True)
        one.reset_stats()
        t0 = time.perf_counter()
        one.steps(a.steps)
        one.sync()
        single_ms = (time.perf_counter() - t0) * 1e3 / a.steps
        s1 = one.stats()
    single = dict(step_ms=single_ms, walk_ms=s1.force_kernel_ms / max(1, s1.force_launches) * (s1.force_launches / a.steps),
                  tree_nodes=int(s1.tree_nodes))
    single["build_and_rest_ms"] = single["step_ms"] - single["walk_ms"]

    sims = [nb.Simulation(ics, *box, method=nb.BARNES_HUT, math_mode=nb.FAST, rank=r, world_size=G, capacity=a.n,
                          shard_mode=nb.SHARD_SPATIAL) for r in range(G)]
    for s in sims:
        s.settings = st
        s.set_balance(a.balance == "work")
        s.init()
    for _ in range(4):
        nb.spatial_step(sims)
    for s in sims:
        s.set_profiling(True)
        s.reset_stats()
    for _ in range(a.steps):
        nb.spatial_step(sims)
    ranks = []
    for r, s in enumerate(sims):
        l = s.let_stats()
        sst = s.stats()
        k = float(l.steps)
        ph = [l.phase_ms[i] / k for i in range(5)]
        ranks.append(dict(rank=r, bodies=len(s), nodes_local=l.nodes_local / k, nodes_global=l.nodes_global / k,
                          nodes_sent=l.nodes_sent / k, nodes_received=l.nodes_received / k, migrated=l.bodies_migrated / k,
                          bytes_sent=l.bytes_sent / k, bytes_allgather=l.bytes_allgather_equivalent / k,
                          node_visits=sst.node_visits / k, accepted=sst.interactions / k,
                          phase_ms=dict(drift_retain_migrate=ph[0], append_keys_sort=ph[1], emit_slice=ph[2], finish_flag_pack=ph[3],
                                        walk_kick=ph[4]),
                          build_ms=ph[1] + ph[2] + ph[3], device_ms=sum(ph)))
    for s in sims:
        s.close()
    out = dict(n=a.n, ranks=G, theta=a.theta, steps=a.steps, single_gpu_device_tree=single, per_rank=ranks,
               worst_rank_device_ms=max(x["device_ms"] for x in ranks),
               bytes_sent_per_rank_per_step=float(np.mean([x["bytes_sent"] for x in ranks])),
               bytes_allgather_per_rank_per_step=float(np.mean([x["bytes_allgather"] for x in ranks])),
               note="one-GPU emulation: phase times are device time of each rank's kernels run alone on the GPU; the exchanges "
                    "(copies here, RCCL in production) are not in them")
    print(f"N = {a.n}, {G} ranks, theta = {a.theta}")
    print(f"single GPU, device tree: {single['step_ms']:.3f} ms/step (walk {single['walk_ms']:.3f}, build + rest {single['build_and_rest_ms']:.3f}), "
          f"{single['tree_nodes']} nodes")
    print(f"{'rank':>4} {'bodies':>8} {'nodes':>8} {'sent':>8} {'recv':>8} {'MB sent':>8} {'MB allg':>8} | "
          f"{'p0':>6} {'p1':>6} {'p2':>6} {'p3':>6} {'walk':>6} {'sum':>6} | {'Mvisits':>8}")
    for x in ranks:
        p = x["phase_ms"]
        print(f"{x['rank']:>4} {x['bodies']:>8} {x['nodes_local']:>8.0f} {x['nodes_sent']:>8.0f} {x['nodes_received']:>8.0f} "
              f"{x['bytes_sent'] / 1e6:>8.2f} {x['bytes_allgather'] / 1e6:>8.2f} | {p['drift_retain_migrate']:>6.3f} {p['append_keys_sort']:>6.3f} "
              f"{p['emit_slice']:>6.3f} {p['finish_flag_pack']:>6.3f} {p['walk_kick']:>6.3f} {x['device_ms']:>6.3f} | {x['node_visits'] / 1e6:>8.1f}")
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
