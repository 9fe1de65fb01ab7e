"""Generates tests/golden/*.npz from the CPU oracle (oracle/nbody_oracle.cpp).

The reference holds no golden vectors for this path and cannot be built
or imported here, so these fixtures are outputs of the RESTATEMENT, pinned by the analytic cases
in tests/test_oracle_pins.py ("parity unpinned" against the reference itself).  They freeze the
oracle: any later edit that changes its rounding sequence fails tests/test_golden.py.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

nb = graft.load_package()
orc = graft.load_oracle()
OUT = os.path.dirname(os.path.abspath(__file__))
CENTER, WIDTH = (0.0, 0.0, 0.0), 64.0


def run(kind, n, ftype, steps, st):
    ics = nb.plummer(n, seed=20250523)
    a = ics.astype(orc.P32) if ftype == "f32" else orc.to_f64(ics)
    counts = []
    for _ in range(steps):
        if kind == "bf":
            a = orc.bf_step_by(a, st, CENTER, WIDTH, st["dt"])
        else:   # "bh": src/manual leaf rule; "bhd": the src/llm walk on the same tree (leaf_mode 1)
            a, acc, vis = orc.bh_step_by(a, st, CENTER, WIDTH, st["dt"], threads=1, leaf_mode=1 if kind == "bhd" else 0)
            counts.append((acc, vis))
    return ics, a, np.array(counts, dtype=np.uint64)


def main():
    cases = []
    for n in (64, 256):
        for ftype in ("f32", "f64"):
            cases.append(("bf", n, ftype, dict(g=1.0, g_soft=0.0, dt=1e-3, theta2=0.5)))
            cases.append(("bh", n, ftype, dict(g=1.0, g_soft=0.0, dt=1e-3, theta2=0.25)))
            cases.append(("bh", n, ftype, dict(g=1.0, g_soft=0.0, dt=1e-3, theta2=0.5)))
            cases.append(("bhd", n, ftype, dict(g=1.0, g_soft=0.01, dt=1e-3, theta2=0.25)))
    only = sys.argv[1] if len(sys.argv) > 1 else None   # e.g. "bhd": write only that kind
    for kind, n, ftype, st in cases:
        if only and kind != only:
            continue
        ics, out, counts = run(kind, n, ftype, 10, st)
        name = f"{kind}_n{n}_{ftype}_t{int(st['theta2'] * 100):03d}.npz"
        np.savez_compressed(os.path.join(OUT, name), kind=kind, n=n, ftype=ftype, steps=10,
                            settings=np.array([st["g"], st["g_soft"], st["dt"], st["theta2"]]),
                            center=np.array(CENTER), width=WIDTH, ics=ics,
                            position=out["position"], velocity=out["velocity"], acceleration=out["acceleration"],
                            mass=out["mass"], counts=counts)
        print("wrote", name)


if __name__ == "__main__":
    main()
