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


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("math", ["strict", "fast"])
@pytest.mark.parametrize("n,split", [(3001, 0), (777, 1), (20000, 4)])
def test_alternative_walk_kernels_same_nodes(gpu, orc, variant, math, n, split):
    # (strict math ignores the switch: it always walks with the parity kernel)
    """The selectable walk kernels -- 1: wave-cooperative (one scalar node load per wave), 2: two lanes
    per body (one contiguous 32-byte request per visit) -- evaluate exactly the opening tests of the
    default one: node counts equal the oracle's, accelerations to rounding.  Odd body counts leave a
    half-filled last lane pair."""
    import ctypes
    nb = gpu
    sd, st = sd_st(nb, theta2=0.25, g_soft=0.01)
    ics = nb.plummer(n, seed=33)
    ref = ics.copy().astype(orc.P32)
    acc_n, vis_n = orc.bh_update_forces(ref, sd, BOX[0], BOX[1], threads=4)
    var = ctypes.c_int.in_dll(nb.lib, "nbody_bh_walk_variant")
    spl = ctypes.c_int.in_dll(nb.lib, "nbody_bh_walk_split")
    var.value, spl.value = variant, split
    try:
        with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.STRICT if math == "strict" else nb.FAST, tree_build=nb.TREE_HOST) as sim:
            sim.settings = st
            sim.update_forces()
            got = sim.get_points()
            s = sim.stats()
    finally:
        var.value, spl.value = 0, 0
    assert (s.interactions, s.node_visits) == (acc_n, vis_n)
    assert rel_err(got["acceleration"], ref["acceleration"]) < 1e-5


@pytest.mark.parametrize("tree", ["host", "device"])
@pytest.mark.parametrize("leaf", ["reference", "direct"])
@pytest.mark.parametrize("n,split,hot,block", [(3001, 0, 2048, 1024), (777, 1, 64, 256), (20000, 4, 1024, 512),
                                              (20000, 8, 4096, 1024), (65536, 0, 2048, 1024), (9, 0, 2048, 1024)])
def test_lds_staged_walk_equals_the_plain_fast_walk(gpu, orc, tree, leaf, n, split, hot, block):
    """Variant 3 (north_star's "cell list staged in LDS"): the most-visited node records live in an LDS table per
    workgroup, the walk follows explicit links instead of pre-order index arithmetic.  It evaluates the same
    opening tests in the same order and adds the same per-segment sums as k_bh_walk: node counts equal the
    oracle's (host tree) and the accelerations equal the plain fast walk's BIT FOR BIT, for every table size
    (smaller than, about, and larger than the number of flagged nodes), workgroup size, split and leaf rule.
    Several steps, so the threshold control and the re-staging of a changed tree are exercised."""
    import ctypes
    nb = gpu
    sd, st = sd_st(nb, theta2=0.25, g_soft=0.01)
    ics = nb.plummer(n, seed=35)
    var = ctypes.c_int.in_dll(nb.lib, "nbody_bh_walk_variant")
    spl = ctypes.c_int.in_dll(nb.lib, "nbody_bh_walk_split")
    cap = ctypes.c_int.in_dll(nb.lib, "nbody_bh_hot_cap")
    blk = ctypes.c_int.in_dll(nb.lib, "nbody_bh_walk_lds_block")
    old = (var.value, spl.value, cap.value, blk.value)
    kw = dict(method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE if tree == "device" else nb.TREE_HOST,
              leaf_mode=nb.LEAF_DIRECT if leaf == "direct" else nb.LEAF_REFERENCE)
    out = {}
    try:
        for v in (0, 3):
            var.value, spl.value, cap.value, blk.value = v, split, hot, block
            with nb.Simulation(ics, *BOX, **kw) as sim:
                sim.settings = st
                sim.update_forces()
                first = sim.get_points()
                s1 = sim.stats()
                sim.steps(4)
                out[v] = (first, (s1.interactions, s1.node_visits), sim.get_points(), sim.stats())
    finally:
        var.value, spl.value, cap.value, blk.value = old
    assert out[3][1] == out[0][1]
    assert np.array_equal(out[3][0]["acceleration"].view(np.uint32), out[0][0]["acceleration"].view(np.uint32))
    assert (out[3][3].interactions, out[3][3].node_visits) == (out[0][3].interactions, out[0][3].node_visits)
    for f in ("position", "velocity", "acceleration"):
        assert np.array_equal(out[3][2][f].view(np.uint32), out[0][2][f].view(np.uint32)), f
    if tree == "host" and leaf == "reference" and n <= 20000:
        ref = ics.copy().astype(orc.P32)
        assert out[3][1] == orc.bh_update_forces(ref, sd, BOX[0], BOX[1], threads=4)
        assert rel_err(out[3][0]["acceleration"], ref["acceleration"]) < 1e-5


@pytest.mark.parametrize("tree", ["host", "device"])
@pytest.mark.parametrize("leaf", ["reference", "direct"])
@pytest.mark.parametrize("n,split", [(3001, 0), (777, 1), (20000, 4), (20000, 24), (65536, 0), (9, 0), (1, 0), (65, 2)])
def test_cooperative_window_walk_equals_the_plain_fast_walk(gpu, orc, tree, leaf, n, split):
    """Variant 4: the 64 lanes of a wave step through the union of their node sequences with a wave-uniform node
    index, node records come from a 64-record LDS window filled by one coalesced load.  Per lane the opening tests,
    their order, the arithmetic and the per-segment sums are those of k_bh_walk: node counts equal the oracle's and
    accelerations and trajectories equal the plain fast walk's BIT FOR BIT (both leaf rules, both tree builds,
    ragged last wave, one body)."""
    import ctypes
    nb = gpu
    sd, st = sd_st(nb, theta2=0.25, g_soft=0.01)
    ics = nb.plummer(n, seed=36)
    var = ctypes.c_int.in_dll(nb.lib, "nbody_bh_walk_variant")
    spl = ctypes.c_int.in_dll(nb.lib, "nbody_bh_walk_split")
    old = (var.value, spl.value)
    kw = dict(method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE if tree == "device" else nb.TREE_HOST,
              leaf_mode=nb.LEAF_DIRECT if leaf == "direct" else nb.LEAF_REFERENCE)
    out = {}
    try:
        for v in (0, 4):
            var.value, spl.value = v, (split if split else 8)   # the automatic split need not be the same for both kernels
            with nb.Simulation(ics, *BOX, **kw) as sim:
                sim.settings = st
                sim.update_forces()
                first = sim.get_points()
                s1 = sim.stats()
                sim.steps(3)
                out[v] = (first, (s1.interactions, s1.node_visits), sim.get_points(), sim.stats())
    finally:
        var.value, spl.value = old
    assert out[4][1] == out[0][1]
    assert np.array_equal(out[4][0]["acceleration"].view(np.uint32), out[0][0]["acceleration"].view(np.uint32))
    assert (out[4][3].interactions, out[4][3].node_visits) == (out[0][3].interactions, out[0][3].node_visits)
    for f in ("position", "velocity", "acceleration"):
        assert np.array_equal(out[4][2][f].view(np.uint32), out[0][2][f].view(np.uint32)), f
    if tree == "host" and leaf == "reference" and n <= 20000:
        ref = ics.copy().astype(orc.P32)
        assert out[4][1] == orc.bh_update_forces(ref, sd, BOX[0], BOX[1], threads=4)
        assert rel_err(out[4][0]["acceleration"], ref["acceleration"]) < 1e-5


@pytest.mark.parametrize("tree", ["host", "device"])
@pytest.mark.parametrize("leaf", ["reference", "direct"])
@pytest.mark.parametrize("n,split", [(3001, 0), (777, 1), (20000, 4), (20000, 24), (65536, 0), (9, 0), (1, 0), (2, 0), (65, 2)])
def test_cooperative_block_walk_same_nodes(gpu, orc, tree, leaf, n, split):
    """Variant 5: a wave pops a block of sibling records from a level-order copy of the tree, tests every child for
    the lanes that opened the parent and pushes the blocks of opened children.  Every lane evaluates exactly the
    opening tests of its own walk: node counts equal the plain fast walk's (and the oracle's on the host tree); the
    accepted monopoles are added in another (fixed) order, so accelerations agree to rounding -- 2e-6 of the largest
    acceleration against the plain walk, 1e-5 against the oracle -- and the result is reproducible run to run."""
    import ctypes
    nb = gpu
    sd, st = sd_st(nb, theta2=0.25, g_soft=0.01)
    ics = nb.plummer(n, seed=37)
    var = ctypes.c_int.in_dll(nb.lib, "nbody_bh_walk_variant")
    spl = ctypes.c_int.in_dll(nb.lib, "nbody_bh_walk_split")
    old = (var.value, spl.value)
    kw = dict(method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE if tree == "device" else nb.TREE_HOST,
              leaf_mode=nb.LEAF_DIRECT if leaf == "direct" else nb.LEAF_REFERENCE)
    out = {}
    try:
        for v in (0, 5, 55):
            var.value, spl.value = v % 50, (split if split else 8)
            with nb.Simulation(ics, *BOX, **kw) as sim:
                sim.settings = st
                sim.update_forces()
                first = sim.get_points()
                s1 = sim.stats()
                sim.steps(3)
                out[v] = (first, (s1.interactions, s1.node_visits), sim.get_points(), sim.stats())
    finally:
        var.value, spl.value = old
    assert out[5][1] == out[0][1]
    assert rel_err(out[5][0]["acceleration"], out[0][0]["acceleration"]) < 2e-6
    assert np.array_equal(out[5][0]["acceleration"].view(np.uint32), out[55][0]["acceleration"].view(np.uint32))   # reproducible
    assert np.array_equal(out[5][2]["position"].view(np.uint32), out[55][2]["position"].view(np.uint32))
    assert rel_err(out[5][2]["position"], out[0][2]["position"]) < 1e-6
    if tree == "host":   # (the device tree's rounding of a centre of mass can flip an opening test that sits on the edge)
        assert (out[5][3].interactions, out[5][3].node_visits) == (out[0][3].interactions, out[0][3].node_visits)
    if tree == "host" and leaf == "reference" and n <= 20000:
        ref = ics.copy().astype(orc.P32)
        assert out[5][1] == orc.bh_update_forces(ref, sd, BOX[0], BOX[1], threads=4)
        assert rel_err(out[5][0]["acceleration"], ref["acceleration"]) < 1e-5


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
