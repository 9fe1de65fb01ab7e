"""Is the device-side tree build reproducible from run to run?  Builds the same tree repeatedly and
compares the exported node arrays bit for bit."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
nb = graft.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ics = nb.plummer(n, seed=3)
outs = []
for rep in range(40):
    with nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE) as sim:
        sim.settings = nb.Settings(1.0, 0.01, 1e-3, 0.25)
        sim.update_forces()
        t = sim.tree()
        outs.append((np.array(t["com_mass"]).copy(), sim.get_points()["acceleration"].copy()))
for rep in range(1, 40):
    a, b = outs[0], outs[rep]
    dn, db = int((a[0] != b[0]).any(axis=1).sum()), int((a[1] != b[1]).any(axis=1).sum())
    if dn or db:
        print(f"run {rep}: nodes differing {dn} of {len(a[0])}, bodies with different acceleration {db}")
print("compared", len(outs), "builds of", n, "bodies")
