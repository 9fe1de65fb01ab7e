"""Parity of the Barnes-Hut HIP path (host octree + K5 walk through the C ABI) against the oracle.
Index/integer work is bit-exact: the octree (cells, pre-order, skip links, centre-of-mass bits)
and the accepted/visited node counts.  Accelerations: the device adds accepted monopoles into one
running sum while the reference nests the sums per tree level, so they agree to rounding:
<= 1e-5 of the largest acceleration."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

BOX = ((0.0, 0.0, 0.0), 64.0)


def sd_st(nb, **kw):
    d = dict(g=1.0, g_soft=0.0, dt=1e-3, theta2=0.5)
    d.update(kw)
    return d, nb.Settings(**d)


@pytest.mark.parametrize("n", [1, 2, 9, 300, 1024, 5000])
@pytest.mark.parametrize("theta2", [0.25, 0.5, 1.0])
def test_update_forces_tree_counts_and_accelerations(gpu, orc, n, theta2):
    nb = gpu
    sd, st = sd_st(nb, theta2=theta2, g_soft=0.02)
    ics = nb.plummer(n, seed=10 + n)
    ref = ics.copy().astype(orc.P32)
    acc_n, vis_n = orc.bh_update_forces(ref, sd, BOX[0], BOX[1], threads=4)
    rt = orc.bh_build_tree(ics.astype(orc.P32), BOX[0], BOX[1])
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.STRICT) as sim:
        sim.settings = st
        sim.update_forces()
        got = sim.get_points()
        s = sim.stats()
        t = sim.tree()
    assert np.array_equal(t["com_mass"].view(np.uint32), rt["com_mass"].view(np.uint32))
    assert np.array_equal(t["width"], rt["width"]) and np.array_equal(t["skip"], rt["skip"])
    assert (s.interactions, s.node_visits, s.tree_nodes) == (acc_n, vis_n, len(rt["width"]))
    if np.abs(ref["acceleration"]).max() > 0:
        assert rel_err(got["acceleration"], ref["acceleration"]) < 1e-5
    else:
        assert not got["acceleration"].any()
    assert np.array_equal(got["position"], ics["position"])


@pytest.mark.parametrize("theta2,expect", [(3.0, (2, 4)), (1.0, (2, 6)), (0.25, (0, 6))])
def test_leaf_drop_cases(gpu, orc, theta2, expect):
    """The two-body cases of tests/test_oracle_pins.py on the device: root accepted with the
    body's own mass inside; other leaf accepted; other leaf rejected -> exactly zero."""
    nb = gpu
    ics = np.zeros(2, nb.PARTICLE_DTYPE)
    ics["position"] = [[-1, -1, -1], [1, 1, 1]]
    ics["mass"] = [1, 3]
    sd, st = sd_st(nb, theta2=theta2)
    ref = ics.copy().astype(orc.P32)
    orc.bh_update_forces(ref, sd, (0, 0, 0), 4.0, 1)
    with nb.Simulation(ics, (0, 0, 0), 4.0, method=nb.BARNES_HUT, math_mode=nb.STRICT) as sim:
        sim.settings = st
        sim.update_forces()
        got = sim.get_points()
        s = sim.stats()
    assert (s.interactions, s.node_visits) == expect
    # one accepted node per body at most: no summation-order freedom, so bit-exact
    assert np.array_equal(got["acceleration"].view(np.uint32), ref["acceleration"].view(np.uint32))


def test_steps_follow_the_oracle(gpu, orc):
    """10 x step_by on 2 000 bodies: counts per step equal, trajectories to rounding."""
    nb = gpu
    sd, st = sd_st(nb, theta2=0.25, g_soft=0.01)
    ics = nb.plummer(2000, seed=31)
    ref = ics.copy().astype(orc.P32)
    tot_a = tot_v = 0
    for _ in range(10):
        ref, a, v = orc.bh_step_by(ref, sd, BOX[0], BOX[1], sd["dt"], threads=4)
        tot_a += a
        tot_v += v
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.STRICT) as sim:
        sim.settings = st
        sim.init()
        sim.steps(10)
        got = sim.get_points()
        s = sim.stats()
        assert sim.elapsed() == pytest.approx(10e-3, rel=1e-5)
    assert (s.interactions, s.node_visits, s.steps) == (tot_a, tot_v, 10)
    assert np.abs(got["position"].astype(np.float64) - ref["position"]).max() < 1e-6
    assert rel_err(got["velocity"], ref["velocity"]) < 1e-5


@pytest.mark.parametrize("n,theta2,g_soft", [(2, 0.5, 0.0), (9, 1.0, 0.0), (300, 0.25, 0.0), (5000, 0.5, 0.02), (20000, 0.25, 0.01), (65536, 0.25, 0.01)])
def test_strict_accelerations_are_bit_exact(gpu, orc, n, theta2, g_soft):
    """Strict math walks with the reference's NESTED sums (k_bh_walk_nested: every opened cell folds its
    children's results left to right from zero, barnes_hut.rs:196-202) on the bit-exact host tree: the
    accelerations equal the oracle's bit for bit, not just to rounding."""
    nb = gpu
    sd, st = sd_st(nb, theta2=theta2, g_soft=g_soft, g=1.25)
    ics = nb.plummer(n, seed=90 + n)
    ref = ics.copy().astype(orc.P32)
    acc_n, vis_n = orc.bh_update_forces(ref, sd, BOX[0], BOX[1], threads=8)
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.STRICT) as sim:
        sim.settings = st
        sim.update_forces()
        got = sim.get_points()
        s = sim.stats()
    assert (s.interactions, s.node_visits) == (acc_n, vis_n)
    assert np.array_equal(got["acceleration"].view(np.uint32), ref["acceleration"].view(np.uint32))


def test_strict_trajectory_is_bit_exact_with_escapes(gpu, orc):
    """... and therefore whole trajectories: positions, velocities, accelerations after 12 steps in a
    tight box (bodies leave on the way), single shard and three shards."""
    nb = gpu
    box = ((0.0, 0.0, 0.0), 3.0)
    sd, st = sd_st(nb, theta2=0.25, g_soft=0.05, dt=1e-2)
    ics = nb.plummer(4000, seed=97)
    ref = ics.copy().astype(orc.P32)
    for _ in range(12):
        ref, _, _ = orc.bh_step_by(ref, sd, box[0], box[1], sd["dt"], threads=4)
    assert len(ref) < 4000
    with nb.Simulation(ics, *box, method=nb.BARNES_HUT, math_mode=nb.STRICT) as sim:
        sim.settings = st
        sim.init()
        sim.steps(12)
        got = sim.get_points()
    sims = [nb.Simulation(ics, *box, method=nb.BARNES_HUT, math_mode=nb.STRICT, rank=r, world_size=3, capacity=len(ics)) for r in range(3)]
    for s_ in sims:
        s_.settings = st
        s_.init()
    for _ in range(12):
        nb.sharded_step(sims)
    got3 = np.concatenate([s_.get_points() for s_ in sims])
    for s_ in sims:
        s_.close()
    for g_ in (got, got3):
        assert len(g_) == len(ref)
        for f in ("position", "velocity", "acceleration", "mass"):
            assert np.array_equal(g_[f].view(np.uint32), ref[f].view(np.uint32)), f


def test_fast_math_walk_same_nodes(gpu, orc):
    """fast math changes only the monopole evaluation (v_rsq_f32); the opening tests are the same."""
    nb = gpu
    sd, st = sd_st(nb, theta2=0.25, g_soft=0.01)
    ics = nb.plummer(3000, seed=32)
    ref = ics.copy().astype(orc.P32)
    acc_n, vis_n = orc.bh_update_forces(ref, sd, BOX[0], BOX[1], threads=4)
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_HOST) as sim:
        sim.settings = st
        sim.update_forces()
        got = sim.get_points()
        s = sim.stats()
    assert (s.interactions, s.node_visits) == (acc_n, vis_n)
    assert rel_err(got["acceleration"], ref["acceleration"]) < 1e-5


@pytest.mark.parametrize("f64", [False, True])
@pytest.mark.parametrize("leaf", ["reference", "direct"])
@pytest.mark.parametrize("n,split", [(3, 1), (1001, 1), (4097, 8), (30000, 16)])
def test_several_bodies_per_lane_walk_exactly_like_one(gpu, orc, n, split, leaf, f64):
    """k_bh_walk_duo / k_bh_walk_fast64<BPL>: a lane walks 2, 3, 4, 6 or 8 neighbouring bodies of the tree order in lockstep
    and fetches the union of their node sequences once.  Per body it evaluates the opening tests of its own walk in its
    own order: with the same node-range segments the accelerations are the one-body-per-lane walk's BIT FOR BIT and the
    accepted / visited counts are equal (and, with the host tree, the oracle's) -- for body counts that leave the last lane
    half empty, both leaf rules, host and device tree, several steps with bodies leaving the box."""
    nb = gpu
    sd, st = sd_st(nb, theta2=0.25, g_soft=0.01)
    ics = nb.plummer(n, seed=37, f64=f64)
    box = ((0.0, 0.0, 0.0), 6.0)
    ics = ics[(np.abs(ics["position"]) < 2.9).all(axis=1)]   # (inside the box to start with; some leave during the steps)
    ics["velocity"] *= 30
    for tree in (nb.TREE_HOST, nb.TREE_DEVICE):
        out = {}
        for bpl in (1, 2, 3, 4, 6, 8):
            with nb.Simulation(ics, *box, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=tree,
                               leaf_mode=nb.LEAF_DIRECT if leaf == "direct" else nb.LEAF_REFERENCE,
                               tuning=dict(bh_walk_duo=bpl, bh_walk_split=split)) as sim:
                sim.settings = st
                sim.update_forces()
                first = sim.get_points()
                s1 = sim.stats()
                sim.steps(3)
                out[bpl] = (first, (s1.interactions, s1.node_visits), sim.get_points(), sim.stats())
        word = np.uint64 if f64 else np.uint32
        for bpl in (2, 3, 4, 6, 8):
            assert out[bpl][1] == out[1][1], (tree, bpl)
            assert np.array_equal(out[bpl][0]["acceleration"].view(word), out[1][0]["acceleration"].view(word)), (tree, bpl)
            assert len(out[bpl][2]) == len(out[1][2])
            for f in ("position", "velocity", "acceleration"):
                assert np.array_equal(out[bpl][2][f].view(word), out[1][2][f].view(word)), (tree, bpl, f)
            assert (out[bpl][3].interactions, out[bpl][3].node_visits) == (out[1][3].interactions, out[1][3].node_visits)
        if tree == nb.TREE_HOST and leaf == "reference" and not f64:
            ref = ics.copy().astype(orc.P32)
            assert out[2][1] == orc.bh_update_forces(ref, sd, box[0], box[1], threads=4)


@pytest.mark.parametrize("f64", [False, True])
def test_xcd_aware_lane_groups_change_nothing_but_the_order_of_work(gpu, f64):
    """bh_walk_xcd (default on): XCD j walks the j-th eighth of the tree order instead of every eighth lane group.  Which
    workgroup walks which bodies does not enter any sum: bit-identical accelerations and equal counts, also when the
    number of lane groups is not a multiple of eight."""
    nb = gpu
    _, st = sd_st(nb, theta2=0.25, g_soft=0.01)
    for n in (30011, 70000):
        ics = nb.plummer(n, seed=38, f64=f64)
        out = []
        for xcd in (0, 1):
            with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE, tuning=dict(bh_walk_xcd=xcd)) as sim:
                sim.settings = st
                sim.steps(2)
                s = sim.stats()
                out.append((sim.get_points(), s.interactions, s.node_visits))
        word = np.uint64 if f64 else np.uint32
        assert out[0][1:] == out[1][1:]
        for f in ("position", "velocity", "acceleration"):
            assert np.array_equal(out[0][0][f].view(word), out[1][0][f].view(word)), (n, f)


def test_retain_in_a_tight_box(gpu, orc):
    nb = gpu
    box = ((0.0, 0.0, 0.0), 2.0)
    sd, st = sd_st(nb, theta2=0.5, g_soft=0.05, dt=2e-2)
    ics = nb.plummer(1500, seed=33)
    ref = ics.copy().astype(orc.P32)
    with nb.Simulation(ics, *box, method=nb.BARNES_HUT, math_mode=nb.STRICT) as sim:
        sim.settings = st
        for _ in range(8):
            sim.step()
            ref, _, _ = orc.bh_step_by(ref, sd, box[0], box[1], sd["dt"], threads=2)
            assert len(sim) == len(ref)
        got = sim.get_points()
    assert len(ref) < 1400
    assert np.array_equal(got["mass"], ref["mass"])
    assert np.abs(got["position"].astype(np.float64) - ref["position"]).max() < 1e-5


def test_reference_workload_disc(gpu, orc):
    """The reference's own configuration (src/main.rs:97-105): disc ICs, box 10, dt=3e-2,
    g_soft=0.02, theta2=1.0."""
    nb = gpu
    box = ((0.0, 0.0, 0.0), 10.0)
    sd, st = sd_st(nb, theta2=1.0, g_soft=0.02, dt=3e-2)
    ics = nb.disc(4000, seed=2)
    ref = ics.copy().astype(orc.P32)
    tot = 0
    for _ in range(5):
        ref, a, _ = orc.bh_step_by(ref, sd, box[0], box[1], sd["dt"], threads=4)
        tot += a
    with nb.Simulation(ics, *box, method=nb.BARNES_HUT, math_mode=nb.STRICT, host_threads=4) as sim:
        sim.settings = st
        sim.init()
        sim.steps(5)
        got = sim.get_points()
        s = sim.stats()
    assert len(got) == len(ref) and s.interactions == tot
    assert np.abs(got["position"].astype(np.float64) - ref["position"]).max() < 1e-5


def test_coincident_bodies_report_tree_depth(gpu):
    nb = gpu
    ics = np.zeros(3, nb.PARTICLE_DTYPE)
    ics["position"] = [[0.3, 0.3, 0.3], [0.3, 0.3, 0.3], [1, 1, 1]]
    ics["mass"] = 1
    with nb.Simulation(ics, (0, 0, 0), 4.0, method=nb.BARNES_HUT) as sim:
        with pytest.raises(nb.NbodyError) as e:
            sim.update_forces()
        assert e.value.code == nb.NBODY_ERR_TREE_DEPTH


def test_clone_drops_tree_and_keeps_state(gpu, orc):
    nb = gpu
    sd, st = sd_st(nb, theta2=0.25)
    ics = nb.plummer(500, seed=34)
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT) as sim:
        sim.settings = st
        sim.steps(2)
        with sim.clone() as twin:
            sim.steps(2)
            twin.steps(2)
            assert np.array_equal(sim.get_points(), twin.get_points())


# ------------------------------------------------------------------ BASELINE.json full size
def test_full_size_65536_theta_half(gpu, orc):
    """configs[2]: 65 536 bodies, theta = 0.5 (theta2 = 0.25).  The threaded oracle finishes one
    force pass in seconds: tree bit-exact, node counts equal, accelerations to rounding."""
    nb = gpu
    sd, st = sd_st(nb, theta2=0.25, g_soft=1e-2)
    ics = nb.plummer(65536)
    ref = ics.copy().astype(orc.P32)
    acc_n, vis_n = orc.bh_update_forces(ref, sd, BOX[0], BOX[1], threads=16)
    rt = orc.bh_build_tree(ics.astype(orc.P32), BOX[0], BOX[1])
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.STRICT) as sim:
        sim.settings = st
        sim.update_forces()
        got = sim.get_points()
        s = sim.stats()
        t = sim.tree()
    assert np.array_equal(t["com_mass"].view(np.uint32), rt["com_mass"].view(np.uint32))
    assert np.array_equal(t["skip"], rt["skip"])
    assert (s.interactions, s.node_visits) == (acc_n, vis_n)
    assert rel_err(got["acceleration"], ref["acceleration"]) < 1e-5


@pytest.mark.parametrize("n", [1, 2, 9, 300, 5000])
@pytest.mark.parametrize("mode", ["f32 strict host tree", "f32 fast device tree", "f64 host tree"])
def test_tree_cells_for_the_renderer_equal_the_references_boxes(gpu, orc, n, mode):
    """nbody_tree_export_cells: what the reference's Barnes-Hut renderer walks (node.bounds.min()/.max() of every node,
    barnes_hut.rs:322-343).  The boxes are recovered top down from the linearised tree with the reference's recurrences and
    equal the oracle's node bounds bit for bit (as f32, which is what the renderer uploads); depths follow the widths."""
    nb = gpu
    f64 = mode.startswith("f64")
    box = ((0.25, -0.5, 0.125), 48.0)
    ics = nb.plummer(n, seed=400 + n, f64=f64)
    ref = ics.copy().astype(orc.P64 if f64 else orc.P32)
    kw = dict(method=nb.BARNES_HUT, math_mode=nb.FAST if "fast" in mode else nb.STRICT,
              tree_build=nb.TREE_DEVICE if "device" in mode else nb.TREE_HOST)
    with nb.Simulation(ics, *box, **kw) as sim:
        sim.settings = nb.Settings(1.0, 0.01, 1e-3, 0.25)
        sim.update_forces()
        mm, depth = sim.tree_cells()
        width = sim.tree()["width"]
    want = orc.bh_tree_cells(ref, box[0], box[1]).astype(np.float32)
    assert mm.shape == want.shape
    assert np.array_equal(mm.view(np.uint32), want.view(np.uint32))
    assert depth[0] == 0 and np.array_equal(depth, np.round(np.log2(box[1] / width.astype(np.float64))).astype(np.int32))
