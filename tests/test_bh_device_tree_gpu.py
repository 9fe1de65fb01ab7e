"""Device-side octree build (NBODY_TREE_DEVICE) against the oracle.
Index work is exact: the same cells in the same pre-order with the same skip links, widths and
leaf bodies as barnes_hut.rs:143-183.  The centre-of-mass sums are f64 prefix sums instead of the
reference's sequential f32 folds: com/mass agree to the rounding of the reference's own fold (n * 2^-24),
so an opening test that sits on the edge can flip: node counts within 1e-3, accelerations <= 1e-5 for
the equal-mass-ish Plummer sets (the disc's 1 : 3e-5 mass ratio makes the reference's f32 root mass
itself 2e-4 off the exact sum)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
BOX = ((0.0, 0.0, 0.0), 64.0)


def build_and_compare(nb, orc, ics, box, sd, math_mode):
    ref = ics.copy().astype(orc.P32)
    acc_n, vis_n = orc.bh_update_forces(ref, sd, box[0], box[1], threads=8)
    rt = orc.bh_build_tree(ics.astype(orc.P32), box[0], box[1])
    with nb.Simulation(ics, *box, method=nb.BARNES_HUT, math_mode=math_mode, tree_build=nb.TREE_DEVICE) as sim:
        sim.settings = nb.Settings(**sd)
        sim.update_forces()
        got = sim.get_points()
        s = sim.stats()
        t = sim.tree()
    assert len(t["skip"]) == len(rt["skip"]) == s.tree_nodes
    assert np.array_equal(t["skip"], rt["skip"]), "skip links differ"
    assert np.array_equal(t["width"], rt["width"]), "cell widths differ"
    leaves = rt["nchild"] == 0
    assert np.array_equal(t["com_mass"][leaves].view(np.uint32), rt["com_mass"][leaves].view(np.uint32)), "leaves are copies of the bodies"
    # a sequential f32 fold over n terms (the reference) carries up to ~n * 2^-24 of rounding; the f64
    # prefix sums are exact to f32: the bound on the difference is the reference's own error
    tol = max(2e-6, len(ics) * 6e-8)
    scale = box[1]
    assert np.abs(t["com_mass"][:, :3].astype(np.float64) - rt["com_mass"][:, :3]).max() < tol * scale
    assert np.allclose(t["com_mass"][:, 3], rt["com_mass"][:, 3], rtol=tol, atol=0)
    return got, ref, s, (acc_n, vis_n)


@pytest.mark.parametrize("n", [1, 2, 3, 9, 300, 1024, 5000, 20000])
@pytest.mark.parametrize("theta2", [0.25, 1.0])
def test_device_tree_structure_and_forces(gpu, orc, n, theta2):
    nb = gpu
    sd = dict(g=1.0, g_soft=0.02, dt=1e-3, theta2=theta2)
    ics = nb.plummer(n, seed=40 + n)
    ics["mass"] *= np.random.default_rng(n).uniform(0.5, 1.5, n).astype(np.float32)
    got, ref, s, (acc_n, vis_n) = build_and_compare(nb, orc, ics, BOX, sd, nb.STRICT)
    assert abs(int(s.interactions) - acc_n) <= max(2, 1e-3 * acc_n)
    assert abs(int(s.node_visits) - vis_n) <= max(2, 1e-3 * vis_n)
    if np.abs(ref["acceleration"]).max() > 0:
        assert rel_err(got["acceleration"], ref["acceleration"]) < 1e-5


def test_device_tree_reference_workload_and_offcentre_box(gpu, orc):
    """The reference's disc in its width-10 box, and a box whose centre arithmetic rounds."""
    nb = gpu
    sd = dict(g=1.0, g_soft=0.02, dt=3e-2, theta2=1.0)
    ics = nb.disc(6000, seed=4)
    build_and_compare(nb, orc, ics, ((0.0, 0.0, 0.0), 10.0), sd, nb.FAST)
    build_and_compare(nb, orc, ics, ((0.1, -0.3, 0.7), 13.7), sd, nb.FAST)


def test_device_tree_steps_track_the_oracle(gpu, orc):
    nb = gpu
    sd = dict(g=1.0, g_soft=0.01, dt=1e-3, theta2=0.25)
    ics = nb.plummer(4000, seed=7)
    ref = ics.copy().astype(orc.P32)
    tot = 0
    for _ in range(10):
        ref, a, _ = orc.bh_step_by(ref, sd, BOX[0], BOX[1], sd["dt"], threads=4)
        tot += a
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE) as sim:
        sim.settings = nb.Settings(**sd)
        sim.init()
        sim.steps(10)
        got = sim.get_points()
        s = sim.stats()
    assert abs(int(s.interactions) - tot) <= 1e-3 * tot
    assert np.abs(got["position"].astype(np.float64) - ref["position"]).max() < 1e-6
    assert rel_err(got["velocity"], ref["velocity"]) < 1e-5


def test_device_tree_with_escapes(gpu, orc):
    nb = gpu
    box = ((0.0, 0.0, 0.0), 2.0)
    sd = dict(g=1.0, g_soft=0.05, dt=2e-2, theta2=0.5)
    ics = nb.plummer(3000, seed=9)
    ref = ics.copy().astype(orc.P32)
    with nb.Simulation(ics, *box, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE) as sim:
        sim.settings = nb.Settings(**sd)
        for _ in range(8):
            sim.step()
            ref, _, _ = orc.bh_step_by(ref, sd, box[0], box[1], sd["dt"], threads=2)
            assert len(sim) == len(ref)
        got = sim.get_points()
    assert len(ref) < 2800 and np.array_equal(got["mass"], ref["mass"])
    assert np.abs(got["position"].astype(np.float64) - ref["position"]).max() < 1e-5


def test_deeper_than_the_device_build_goes_falls_back_to_the_host_build(gpu, orc):
    """Two bodies 2e-7 apart in a width-64 box separate only below level 21.  With the device build's second keys
    switched off (nbody_tree_max_tie = 1) it reports "too deep" and the step uses the host build (exact counts);
    coincident bodies raise either way."""
    nb = gpu
    sd = dict(g=1.0, g_soft=0.01, dt=1e-3, theta2=0.25)
    ics = nb.plummer(500, seed=3)
    ics["position"][7] = ics["position"][3] + np.float32(2e-7)
    ref = ics.copy().astype(orc.P32)
    acc_n, vis_n = orc.bh_update_forces(ref, sd, BOX[0], BOX[1], threads=2)
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.STRICT, tree_build=nb.TREE_DEVICE, tuning=dict(tree_max_tie=1)) as sim:
        sim.settings = nb.Settings(**sd)
        sim.update_forces()
        s = sim.stats()
        got = sim.get_points()
    assert (s.interactions, s.node_visits) == (acc_n, vis_n)
    assert np.array_equal(got["acceleration"].view(np.uint32), ref["acceleration"].view(np.uint32))
    ics["position"][7] = ics["position"][3]
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, tree_build=nb.TREE_DEVICE) as sim:
        with pytest.raises(nb.NbodyError) as e:
            sim.update_forces()
        assert e.value.code == nb.NBODY_ERR_TREE_DEPTH


@pytest.mark.parametrize("n", [2, 500, 6000])
def test_bodies_that_share_all_21_levels_get_second_keys_on_the_device(gpu, orc, n):
    """Pairs, a triple and a clump of 9 bodies a few 1e-7 apart (cells of level 21 are 3e-5 wide; at N = 2^22 a Plummer
    sphere has such a pair at almost every step): the device build orders each group of equal keys by the levels 21..41
    and emits the chain of cells below level 21 -- the same nodes, skip links and widths as the reference's recursion."""
    nb = gpu
    sd = dict(g=1.0, g_soft=0.01, dt=1e-3, theta2=0.25)
    ics = nb.plummer(n, seed=3)
    rng = np.random.default_rng(n)
    def near(dst, src, k=1):
        ics["position"][dst] = ics["position"][src] + (rng.integers(1, 12, 3) * k).astype(np.float32) * np.float32(1.2e-7)
    near(1, 0)
    if n > 2:
        near(7, 3); near(11, 3, 2); near(n - 1, n // 2)
        for q in range(8):
            near(100 + q, 99, q + 1)
    got, ref, s, (acc_n, vis_n) = build_and_compare(nb, orc, ics, BOX, sd, nb.STRICT)
    t_depth = np.log2(BOX[1] / np.float64(orc.bh_build_tree(ics.astype(orc.P32), BOX[0], BOX[1])["width"].min()))
    assert t_depth > 22                                   # the case does go below the first keys
    assert abs(int(s.interactions) - acc_n) <= max(2, 1e-3 * acc_n) and abs(int(s.node_visits) - vis_n) <= max(2, 1e-3 * vis_n)
    # inside a clump the reference's own f32 centres of mass are rounding noise at the scale of the separations, so the
    # odd opening test between clump members falls the other way (and a dropped near-field leaf is a whole pair force)
    clump = np.zeros(n, bool)
    clump[[0, 1] + ([3, 7, 11, n - 1, n // 2] + list(range(99, 108)) if n > 2 else [])] = True
    scale = np.abs(ref["acceleration"]).max() or 1.0   # (n = 2: the reference drops the near-field leaf, no force at all)
    def check(acc):
        err = np.abs(acc.astype(np.float64) - ref["acceleration"]).max(axis=1) / scale
        rest = err[~clump]
        assert np.count_nonzero(rest > 1e-5) <= 2 and rest.max(initial=0.0) < 1e-3 and err[clump].max() < 1e-2, (rest.max(initial=0.0), err[clump].max())
    check(got["acceleration"])
    # the asynchronous fast-math path and a split walk (ancestor lists of the segments run through the deep cells)
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE) as sim:
        sim.settings = nb.Settings(**sd)
        sim.update_forces()
        fast = sim.get_points()
        sf = sim.stats()
    assert sf.tree_nodes == s.tree_nodes
    check(fast["acceleration"])


def test_device_tree_full_size_65536(gpu, orc):
    nb = gpu
    sd = dict(g=1.0, g_soft=1e-2, dt=1e-3, theta2=0.25)
    ics = nb.plummer(65536)
    got, ref, s, (acc_n, vis_n) = build_and_compare(nb, orc, ics, BOX, sd, nb.FAST)
    assert abs(int(s.interactions) - acc_n) <= 1e-3 * acc_n
    assert rel_err(got["acceleration"], ref["acceleration"]) < 1e-5


def test_device_tree_trajectory_stays_with_the_host_tree_trajectory(gpu):
    """What the device build's different rounding of the centres of mass (f64 prefix sums instead of the reference's
    sequential f32 folds) costs over a trajectory: 100 steps of configs[2] (65 536 bodies, theta = 0.5, fast math) with
    the tree built on the device against the same run with the host build.  Per pass the accepted-node counts agree to
    1e-3; after 100 steps the positions differ by a few 1e-7 (the bodies sit at radii ~1, float32 resolution 6e-8), the
    energies by < 1e-6 relative."""
    nb = gpu
    st = nb.Settings(1.0, 1e-2, 1e-3, 0.25)
    ics = nb.plummer(65536)
    out = {}
    for name, tb in (("host", nb.TREE_HOST), ("device", nb.TREE_DEVICE)):
        with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=tb) as sim:
            sim.settings = st
            sim.init()
            sim.steps(100)
            out[name] = (sim.get_points(), sim.stats(), sim.energy())
    a, b = out["host"], out["device"]
    dpos = np.abs(a[0]["position"].astype(np.float64) - b[0]["position"]).max()
    dvel = np.abs(a[0]["velocity"].astype(np.float64) - b[0]["velocity"]).max()
    ea, eb = sum(a[2]), sum(b[2])
    print(f"device vs host tree after 100 steps: max |dpos| {dpos:.3e}, max |dvel| {dvel:.3e}, accepted {b[1].interactions} vs {a[1].interactions}, "
          f"E {eb:.9f} vs {ea:.9f}")
    assert len(a[0]) == len(b[0]) == 65536
    assert abs(int(a[1].interactions) - int(b[1].interactions)) < 1e-3 * a[1].interactions
    assert dpos < 2e-5 and dvel < 2e-3
    assert abs(ea - eb) < 1e-6 * abs(ea)


@pytest.mark.parametrize("clump", [600, 3000, 6000])
def test_a_clump_inside_one_level_16_cell(gpu, orc, clump):
    """The device build's radix sort covers the top 16 levels of the keys; bodies that share them are finished per group --
    a pair by one thread, a clump of hundreds by a workgroup (sort_big_group inside k_tree_ties, up to 4096), and beyond that the build says
    so with a flag of its own and the single-GPU step builds on the host.  Either way the tree is the reference's: node
    counts and accelerations equal the oracle's (strict walk: bit for bit)."""
    nb = gpu
    n = 3000
    rng = np.random.default_rng(clump)
    ics = nb.plummer(n + clump, seed=91)
    w16 = np.float64(BOX[1]) / 65536.0
    lo = -32.0 + 36000 * w16                                            # the lower wall of one level-16 cell
    ics["position"][n:] = (lo + w16 * (0.1 + 0.5 * rng.random((clump, 3)))).astype(np.float32)
    ics["velocity"][n:] = 0.0
    sd = dict(g=1.0, g_soft=1e-3, dt=1e-3, theta2=0.25)
    ref = ics.copy().astype(orc.P32)
    acc_n, vis_n = orc.bh_update_forces(ref, sd, BOX[0], BOX[1], threads=8)
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.STRICT, tree_build=nb.TREE_DEVICE) as sim:
        sim.settings = nb.Settings(**sd)
        sim.update_forces()
        s = sim.stats()
        got = sim.get_points()
        skip = sim.tree()["skip"]
    tree = orc.bh_build_tree(ref, BOX[0], BOX[1])
    assert np.array_equal(skip, tree["skip"])                            # the reference's cells and links
    if clump > 4096:     # built on the host: exact
        assert (s.interactions, s.node_visits) == (acc_n, vis_n)
        assert np.array_equal(got["acceleration"].view(np.uint32), ref["acceleration"].view(np.uint32))
    else:                # built on the device: centres of mass from f64 prefix sums.  The reference folds them in f32, which inside
        # the clump (bodies 1e-4 apart at |x| ~ 3: an f32 ulp is 2.4e-7) is itself good to a per cent of a separation only:
        # there the two differ by the REFERENCE's rounding, so the accelerations are compared where they are well defined
        assert abs(s.interactions - acc_n) <= 1e-3 * acc_n and abs(s.node_visits - vis_n) <= 1e-3 * vis_n
        assert rel_err(got["acceleration"][:n], ref["acceleration"][:n]) < 1e-4   # (the clump is half the mass: its centre, as the reference rounds it, moves every body a little)
        assert rel_err(got["acceleration"], ref["acceleration"]) < 5e-2
