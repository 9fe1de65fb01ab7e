#!/usr/bin/env python3
"""Randomised stress of the shared-fetch walk (k_bh_walk_duo / k_bh_walk_fast64<BPL>): random body counts, boxes, theta,
clustered and degenerate initial conditions (tools/let_stress.py's generator), both leaf rules, both tree builds, f32 and
f64, a forced number of bodies per lane and of node-range segments -- against the one-body-per-lane walk with the same
segments: accelerations and several steps' positions BIT FOR BIT, visited / accepted counts equal (and, host tree, f32,
reference leaf rule: equal to the CPU oracle's).  Exits non-zero at the first mismatch.

    python tools/bh_duo_stress.py [--cases 300] [--seed 1] [--max-n 60000]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402
from let_stress import make_ics  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-n", type=int, default=60000)
    a = ap.parse_args()
    nb = graft.load_package()
    orc = graft.load_oracle()
    rng = np.random.default_rng(a.seed)
    kinds = ["plummer", "two_clumps", "line", "plane", "corner", "near_pairs"]
    t0 = time.time()
    done = 0
    for case in range(a.cases):
        n = int(rng.choice([int(rng.integers(1, 70)), int(rng.integers(70, 3000)), int(rng.integers(3000, a.max_n))]))
        kind = kinds[int(rng.integers(0, len(kinds)))]
        width = float(rng.choice([64.0, 8.0, 3.0]))
        theta2 = float(rng.choice([0.25, 0.49, 1.0, 0.04]))
        leaf = nb.LEAF_DIRECT if rng.random() < 0.4 else nb.LEAF_REFERENCE
        tree = nb.TREE_HOST if rng.random() < 0.4 else nb.TREE_DEVICE
        f64 = bool(rng.random() < 0.3)
        bpl = int(rng.choice([2, 3, 4, 6, 8]))
        split = int(rng.choice([1, 2, 5, 16, 64]))
        xcd = int(rng.integers(0, 2))
        steps = int(rng.integers(0, 4))
        dt = float(rng.choice([1e-3, 2e-2]))
        ics = make_ics(nb, rng, n, kind)
        ics = ics[(np.abs(ics["position"]) <= np.float32(width / 2)).all(axis=1)]
        if len(ics) == 0:
            continue
        if f64:
            wide = np.zeros(len(ics), nb.PARTICLE_DTYPE64)
            for f in ("position", "velocity", "acceleration", "mass"):
                wide[f] = ics[f]
            ics = wide
        n = len(ics)
        box = ((0.0, 0.0, 0.0), width)
        st = nb.Settings(1.0, 0.02, dt, theta2)
        tag = (f"case {case}: n={n} {kind} width={width} theta2={theta2} leaf={'direct' if leaf == nb.LEAF_DIRECT else 'ref'} "
               f"tree={'host' if tree == nb.TREE_HOST else 'device'} {'f64' if f64 else 'f32'} bodies/lane={bpl} segments={split} xcd={xcd} steps={steps}")
        out = []
        try:
            for b in (1, bpl):
                with nb.Simulation(ics, *box, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=tree, leaf_mode=leaf,
                                   tuning=dict(bh_walk_duo=b, bh_walk_split=split, bh_walk_xcd=xcd)) as sim:
                    sim.settings = st
                    sim.update_forces()
                    first = sim.get_points()
                    s1 = sim.stats()
                    if steps:
                        sim.steps(steps)
                    out.append((first, (s1.interactions, s1.node_visits), sim.get_points(), sim.stats()))
        except nb.NbodyError as e:
            print(tag, "-> refused:", e, flush=True)
            continue
        word = np.uint64 if f64 else np.uint32
        ok = out[0][1] == out[1][1] and len(out[0][2]) == len(out[1][2])
        ok = ok and np.array_equal(out[0][0]["acceleration"].view(word), out[1][0]["acceleration"].view(word))
        for f in ("position", "velocity", "acceleration"):
            ok = ok and np.array_equal(out[0][2][f].view(word), out[1][2][f].view(word))
        ok = ok and (out[0][3].interactions, out[0][3].node_visits) == (out[1][3].interactions, out[1][3].node_visits)
        if ok and tree == nb.TREE_HOST and not f64 and n <= 20000:
            sd = dict(g=1.0, g_soft=0.02, dt=dt, theta2=theta2)
            ref = ics.copy().astype(orc.P32)
            want = orc.bh_update_forces(ref, sd, box[0], box[1], threads=8, leaf_mode=1 if leaf == nb.LEAF_DIRECT else 0)
            ok = out[1][1] == want
        print(tag, "-> OK" if ok else "-> MISMATCH", flush=True)
        if not ok:
            sys.exit(1)
        done += 1
    print(f"{done} cases in {time.time() - t0:.0f} s: all OK")


if __name__ == "__main__":
    main()
