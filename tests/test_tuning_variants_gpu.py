"""The experimental Barnes-Hut walks (csrc/kernels_bh.hip, variants 1-5: wave-cooperative, two lanes per body, hot records in
LDS, cooperative window, cooperative block walk).  They were built, measured and lost against the per-lane walk (DESIGN.md
section 3.4); the release library does not carry them.  They stay reproducible in the TUNING build of the same sources
(libnbody_hip_tuning.so, -DNBODY_TUNING), which these tests load beside the product, and must still visit the reference's
nodes: counts against the oracle, accelerations against the default walk of the same build."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

BOX = ((0.0, 0.0, 0.0), 64.0)


def sd_st(nb, **kw):
    d = dict(g=1.0, g_soft=0.0, dt=1e-3, theta2=0.5)
    d.update(kw)
    return d, nb.Settings(**d)


def test_release_library_refuses_the_experimental_knobs(gpu):
    nb = gpu
    assert not nb.is_tuning_build()
    with nb.Simulation(nb.plummer(64), *BOX, method=nb.BARNES_HUT, math_mode=nb.FAST) as sim:
        for name, value in (("bh_walk_variant", 3), ("bh_walk_debug", 1), ("sym_debug", 5)):
            with pytest.raises(nb.NbodyError) as e:
                sim.set_tuning(name, value)
            assert e.value.code == nb.NBODY_ERR_INVALID and "tuning build" in str(e.value)
        sim.set_tuning("bh_walk_variant", 0)       # (the default is always accepted)
        sim.set_tuning("bh_walk_split", 4)
        assert sim.get_tuning("bh_walk_split") == 4
        with pytest.raises(nb.NbodyError):
            sim.set_tuning("no_such_knob", 1)


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("math", ["strict", "fast"])
@pytest.mark.parametrize("n,split", [(3001, 0), (777, 1), (20000, 4)])
def test_alternative_walk_kernels_same_nodes(gpu_tuning, orc, variant, math, n, split):
    # (strict math ignores the switch: it always walks with the parity kernel)
    """The selectable walk kernels -- 1: wave-cooperative (one scalar node load per wave), 2: two lanes
    per body (one contiguous 32-byte request per visit) -- evaluate exactly the opening tests of the
    default one: node counts equal the oracle's, accelerations to rounding.  Odd body counts leave a
    half-filled last lane pair."""
    nb = gpu_tuning
    sd, st = sd_st(nb, theta2=0.25, g_soft=0.01)
    ics = nb.plummer(n, seed=33)
    ref = ics.copy().astype(orc.P32)
    acc_n, vis_n = orc.bh_update_forces(ref, sd, BOX[0], BOX[1], threads=4)
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.STRICT if math == "strict" else nb.FAST, tree_build=nb.TREE_HOST,
                       tuning=dict(bh_walk_variant=variant, bh_walk_split=split)) as sim:
        sim.settings = st
        sim.update_forces()
        got = sim.get_points()
        s = sim.stats()
    assert (s.interactions, s.node_visits) == (acc_n, vis_n)
    assert rel_err(got["acceleration"], ref["acceleration"]) < 1e-5


@pytest.mark.parametrize("tree", ["host", "device"])
@pytest.mark.parametrize("leaf", ["reference", "direct"])
@pytest.mark.parametrize("n,split,hot,block", [(3001, 0, 2048, 1024), (777, 1, 64, 256), (20000, 4, 1024, 512),
                                              (20000, 8, 4096, 1024), (65536, 0, 2048, 1024), (9, 0, 2048, 1024)])
def test_lds_staged_walk_equals_the_plain_fast_walk(gpu_tuning, orc, tree, leaf, n, split, hot, block):
    """Variant 3 (north_star's "cell list staged in LDS"): the most-visited node records live in an LDS table per
    workgroup, the walk follows explicit links instead of pre-order index arithmetic.  It evaluates the same
    opening tests in the same order and adds the same per-segment sums as k_bh_walk: node counts equal the
    oracle's (host tree) and the accelerations equal the plain fast walk's BIT FOR BIT, for every table size
    (smaller than, about, and larger than the number of flagged nodes), workgroup size, split and leaf rule.
    Several steps, so the threshold control and the re-staging of a changed tree are exercised."""
    nb = gpu_tuning
    sd, st = sd_st(nb, theta2=0.25, g_soft=0.01)
    ics = nb.plummer(n, seed=35)
    kw = dict(method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE if tree == "device" else nb.TREE_HOST,
              leaf_mode=nb.LEAF_DIRECT if leaf == "direct" else nb.LEAF_REFERENCE)
    out = {}
    for v in (0, 3):
        with nb.Simulation(ics, *BOX, tuning=dict(bh_walk_variant=v, bh_walk_split=split, bh_hot_cap=hot, bh_walk_lds_block=block), **kw) as sim:
            sim.settings = st
            sim.update_forces()
            first = sim.get_points()
            s1 = sim.stats()
            sim.steps(4)
            out[v] = (first, (s1.interactions, s1.node_visits), sim.get_points(), sim.stats())
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
def test_cooperative_window_walk_equals_the_plain_fast_walk(gpu_tuning, orc, tree, leaf, n, split):
    """Variant 4: the 64 lanes of a wave step through the union of their node sequences with a wave-uniform node
    index, node records come from a 64-record LDS window filled by one coalesced load.  Per lane the opening tests,
    their order, the arithmetic and the per-segment sums are those of k_bh_walk: node counts equal the oracle's and
    accelerations and trajectories equal the plain fast walk's BIT FOR BIT (both leaf rules, both tree builds,
    ragged last wave, one body)."""
    nb = gpu_tuning
    sd, st = sd_st(nb, theta2=0.25, g_soft=0.01)
    ics = nb.plummer(n, seed=36)
    kw = dict(method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE if tree == "device" else nb.TREE_HOST,
              leaf_mode=nb.LEAF_DIRECT if leaf == "direct" else nb.LEAF_REFERENCE)
    out = {}
    for v in (0, 4):
        # (the automatic split need not be the same for both kernels: pinned)
        with nb.Simulation(ics, *BOX, tuning=dict(bh_walk_variant=v, bh_walk_split=split if split else 8), **kw) as sim:
            sim.settings = st
            sim.update_forces()
            first = sim.get_points()
            s1 = sim.stats()
            sim.steps(3)
            out[v] = (first, (s1.interactions, s1.node_visits), sim.get_points(), sim.stats())
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
def test_cooperative_block_walk_same_nodes(gpu_tuning, orc, tree, leaf, n, split):
    """Variant 5: a wave pops a block of sibling records from a level-order copy of the tree, tests every child for
    the lanes that opened the parent and pushes the blocks of opened children.  Every lane evaluates exactly the
    opening tests of its own walk: node counts equal the plain fast walk's (and the oracle's on the host tree); the
    accepted monopoles are added in another (fixed) order, so accelerations agree to rounding -- 2e-6 of the largest
    acceleration against the plain walk, 1e-5 against the oracle -- and the result is reproducible run to run."""
    nb = gpu_tuning
    sd, st = sd_st(nb, theta2=0.25, g_soft=0.01)
    ics = nb.plummer(n, seed=37)
    kw = dict(method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE if tree == "device" else nb.TREE_HOST,
              leaf_mode=nb.LEAF_DIRECT if leaf == "direct" else nb.LEAF_REFERENCE)
    out = {}
    for v in (0, 5, 55):
        with nb.Simulation(ics, *BOX, tuning=dict(bh_walk_variant=v % 50, bh_walk_split=split if split else 8), **kw) as sim:
            sim.settings = st
            sim.update_forces()
            first = sim.get_points()
            s1 = sim.stats()
            sim.steps(3)
            out[v] = (first, (s1.interactions, s1.node_visits), sim.get_points(), sim.stats())
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
