"""F = f64 across the boundary (NbodyConfig.dtype = NBODY_F64): the reference's trait is generic over Float
(src/shared.rs:12-44) and its own driver runs f64 (src/main.rs:52-105).  Strict arithmetic, host-built tree: bit-exact against
the oracle's f64 instantiation (over index-block ranks too: tests/test_multiproc_gpu.py); device-built tree and the fast
walk: to f64 rounding."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
BOX = ((0.0, 0.0, 0.0), 64.0)
FIELDS = ("position", "velocity", "acceleration", "mass")


def eq(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint64), np.ascontiguousarray(b).view(np.uint64))


@pytest.mark.parametrize("n", [1, 2, 65, 1000, 5000])
def test_brute_force_f64_steps_are_bit_exact(gpu, orc, n):
    nb = gpu
    sd = dict(g=1.0, g_soft=0.0, dt=1e-3, theta2=0.5)
    ics = nb.plummer(n, seed=51, f64=True)             # unrounded f64 initial conditions
    assert ics.dtype == nb.PARTICLE_DTYPE64 and ics["position"].dtype == np.float64
    ref = ics.copy().astype(orc.P64)
    steps = 3 if n > 1000 else 6
    for _ in range(steps):
        ref = orc.bf_step_by(ref, sd, BOX[0], BOX[1], sd["dt"])
    with nb.Simulation(ics, *BOX, method=nb.BRUTE_FORCE) as sim:
        assert sim.f64
        sim.settings = nb.Settings(**sd)
        sim.init()
        sim.steps(steps)
        got = sim.get_points()
        s = sim.stats()
        assert sim.elapsed() == sum([sd["dt"]] * steps, 0.0)
    assert s.interactions == steps * n * (n - 1)
    for f in FIELDS:
        assert eq(got[f], ref[f]), f


def test_brute_force_f64_with_escapes_settings_and_negative_dt(gpu, orc):
    """Bodies leave a tight box (retain in f64: inclusive walls), settings change between steps, dt goes negative
    (src/vis.rs:236-251 rewinds with step_by(-dt))."""
    nb = gpu
    box = ((0.1, -0.05, 0.0), 1.7)
    ics = nb.plummer(1500, seed=52, f64=True)
    ref = ics.copy().astype(orc.P64)
    with nb.Simulation(ics, *box, method=nb.BRUTE_FORCE) as sim:
        sim.init()
        for k in range(8):
            sd = dict(g=1.0 + 0.1 * k, g_soft=0.05, dt=2e-2, theta2=0.5)
            sim.settings = nb.Settings(**sd)
            dt = -1e-2 if k == 5 else sd["dt"]
            sim.step_by(dt)
            ref = orc.bf_step_by(ref, sd, box[0], box[1], dt)
            assert len(sim) == len(ref)
        got = sim.get_points()
    assert len(ref) < 1400
    for f in FIELDS:
        assert eq(got[f], ref[f]), f


@pytest.mark.parametrize("leaf", ["reference", "direct"])
@pytest.mark.parametrize("n,theta2", [(1, 0.25), (2, 0.25), (9, 1.0), (1000, 0.25), (5000, 0.5), (20000, 0.25)])
def test_barnes_hut_f64_counts_tree_and_accelerations(gpu, orc, n, theta2, leaf):
    nb = gpu
    sd = dict(g=1.0, g_soft=0.01, dt=1e-3, theta2=theta2)
    ics = nb.plummer(n, seed=53, f64=True)
    ref = ics.copy().astype(orc.P64)
    acc_n, vis_n = orc.bh_update_forces(ref, sd, BOX[0], BOX[1], threads=4, leaf_mode=1 if leaf == "direct" else 0)
    rt = orc.bh_build_tree(ics.astype(orc.P64), BOX[0], BOX[1])
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, leaf_mode=nb.LEAF_DIRECT if leaf == "direct" else nb.LEAF_REFERENCE) as sim:
        sim.settings = nb.Settings(**sd)
        sim.update_forces()
        got = sim.get_points()
        s = sim.stats()
        t = sim.tree()
    assert (s.interactions, s.node_visits) == (acc_n, vis_n)
    assert t["com_mass"].dtype == np.float64 and eq(t["com_mass"], rt["com_mass"]) and np.array_equal(t["skip"], rt["skip"])
    assert eq(t["width"], rt["width"])
    assert eq(got["acceleration"], ref["acceleration"])


@pytest.mark.parametrize("leaf", ["reference", "direct"])
def test_barnes_hut_f64_trajectory_with_escapes(gpu, orc, leaf):
    nb = gpu
    box = ((0.0, 0.0, 0.0), 2.0)
    sd = dict(g=1.0, g_soft=0.05, dt=2e-2, theta2=0.5)
    lm = 1 if leaf == "direct" else 0
    ics = nb.plummer(3000, seed=54, f64=True)
    ref = ics.copy().astype(orc.P64)
    tot_a = tot_v = 0
    with nb.Simulation(ics, *box, method=nb.BARNES_HUT, leaf_mode=nb.LEAF_DIRECT if lm else nb.LEAF_REFERENCE) as sim:
        sim.settings = nb.Settings(**sd)
        sim.init()
        for _ in range(8):
            sim.step()
            ref, a, v = orc.bh_step_by(ref, sd, box[0], box[1], sd["dt"], threads=2, leaf_mode=lm)
            tot_a += a
            tot_v += v
            assert len(sim) == len(ref)
        got = sim.get_points()
        s = sim.stats()
    assert len(ref) < 2800 and (s.interactions, s.node_visits) == (tot_a, tot_v)
    for f in FIELDS:
        assert eq(got[f], ref[f]), f


def test_reference_driver_configuration_in_f64(gpu, orc):
    """What src/main.rs runs: disc ICs in f64, box 10, dt = 3e-2, g_soft = 0.02, theta2 = 1.0, Barnes-Hut."""
    nb = gpu
    box = ((0.0, 0.0, 0.0), 10.0)
    sd = dict(g=1.0, g_soft=0.02, dt=3e-2, theta2=1.0)
    ics = nb.disc(4000, seed=2, f64=True)
    ref = ics.copy().astype(orc.P64)
    tot = 0
    for _ in range(5):
        ref, a, _ = orc.bh_step_by(ref, sd, box[0], box[1], sd["dt"], threads=4)
        tot += a
    with nb.Simulation(ics, *box, method=nb.BARNES_HUT, host_threads=4) as sim:
        sim.settings = nb.Settings(**sd)
        sim.init()
        sim.steps(5)
        got = sim.get_points()
        s = sim.stats()
    assert len(got) == len(ref) and s.interactions == tot
    for f in FIELDS:
        assert eq(got[f], ref[f]), f


def test_f64_add_remove_clone_energy(gpu, orc):
    nb = gpu
    sd = dict(g=1.0, g_soft=0.01, dt=1e-3, theta2=0.5)
    ics = nb.plummer(300, seed=55, f64=True)
    extra = nb.plummer(3, seed=56, f64=True)
    with nb.Simulation(ics, *BOX, method=nb.BRUTE_FORCE, capacity=400) as sim:
        sim.settings = nb.Settings(**sd)
        sim.add_point(extra[0])                       # Vec::push
        sim.remove_point(5)                           # Vec::swap_remove
        ref = np.concatenate([ics, extra[:1]]).astype(orc.P64)
        ref[5] = ref[-1]
        ref = ref[:-1].copy()
        assert len(sim) == 300
        with sim.clone() as twin:
            assert twin.f64
            sim.steps(3)
            twin.steps(3)
            assert np.array_equal(sim.get_points(), twin.get_points())
        for _ in range(3):
            ref = orc.bf_step_by(ref, sd, BOX[0], BOX[1], sd["dt"])
        got = sim.get_points()
        for f in FIELDS:
            assert eq(got[f], ref[f]), f
        ke, pe = sim.energy()
        oke, ope = orc.energy(ref, g=sd["g"], g_soft=sd["g_soft"])
        assert abs(ke - oke) < 1e-12 * abs(oke) and abs(pe - ope) < 1e-12 * abs(ope)
        with pytest.raises(nb.NbodyError):
            sim.remove_point(300)


def test_f32_entry_points_on_an_f64_handle_and_back(gpu):
    """nbody_set_settings / nbody_step_by / nbody_elapsed (f32) widen exactly on an f64 handle; the _f64 forms round on an
    f32 handle.  f64 handles refuse what they do not support."""
    import ctypes as C
    nb = gpu
    ics = nb.plummer(64, seed=57, f64=True)
    with nb.Simulation(ics, *BOX, method=nb.BRUTE_FORCE) as sim:
        sim._check(nb.lib.nbody_set_settings(sim._h, 1.0, 0.25, 0.5, 0.125))
        assert sim.settings == nb.Settings(1.0, 0.25, 0.5, 0.125)
        sim._check(nb.lib.nbody_step_by(sim._h, 0.5))
        assert sim.elapsed() == 0.5
        v = C.c_float()
        sim._check(nb.lib.nbody_elapsed(sim._h, C.byref(v)))
        assert v.value == 0.5
        sim.comm_init(nb.comm_local_id())            # (a world of one with a communicator: f64 worlds shard by index blocks)
        sim.step()
        assert sim.comm_transport() == "ipc" and sim.count_global() == 64
        bad = np.zeros(64, nb.PARTICLE_DTYPE)
        assert nb.lib.nbody_upload(sim._h, bad.ctypes.data, 64, 40) == nb.NBODY_ERR_INVALID      # 40-byte records on an f64 handle
    with nb.Simulation(nb.plummer(64, seed=57), *BOX, method=nb.BRUTE_FORCE) as sim:
        assert not sim.f64
        sim._check(nb.lib.nbody_set_settings_f64(sim._h, 1.0, 0.1, 1e-3, 0.3))
        s = sim.settings
        assert s.dt == float(np.float32(1e-3)) and s.theta2 == float(np.float32(0.3))
        d = C.c_double()
        sim._check(nb.lib.nbody_step_by_f64(sim._h, 1e-3))
        sim._check(nb.lib.nbody_elapsed_f64(sim._h, C.byref(d)))
        assert d.value == float(np.float32(1e-3))
    with pytest.raises(nb.NbodyError) as e:          # f64 worlds are sharded by index blocks only
        nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.FAST, rank=0, world_size=2, capacity=64, shard_mode=nb.SHARD_SPATIAL)
    assert e.value.code == nb.NBODY_ERR_INVALID
    with nb.Simulation(ics, *BOX, rank=1, world_size=2, capacity=64) as half:      # the second block of two
        assert half.f64 and half.local_range() == (32, 32) and len(half) == 32
        with pytest.raises(nb.NbodyError) as e:
            half.step()                                                               # no communicator
        assert e.value.code == nb.NBODY_ERR_COMM
        for call in (lambda: half.add_point(ics[0]), lambda: half.remove_point(0), half.energy):
            with pytest.raises(nb.NbodyError):
                call()


def test_f64_tracks_f32(gpu):
    """Sanity across precisions: the f32 strict run stays within float32 rounding of the f64 run of the same (f32) ICs."""
    nb = gpu
    st = nb.Settings(1.0, 0.01, 1e-3, 0.25)
    ics32 = nb.plummer(2000, seed=58)
    ics64 = np.zeros(2000, nb.PARTICLE_DTYPE64)
    for f in FIELDS:
        ics64[f] = ics32[f]
    out = []
    for ics in (ics32, ics64):
        with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.STRICT) as sim:
            sim.settings = st
            sim.init()
            sim.steps(10)
            out.append(sim.get_points())
    assert rel_err(out[0]["acceleration"], out[1]["acceleration"]) < 1e-5
    assert np.abs(out[0]["position"].astype(np.float64) - out[1]["position"]).max() < 3e-5   # (bodies out to radius 10: f32 ulp 1e-6, 10 steps)


@pytest.mark.parametrize("leaf", ["reference", "direct"])
@pytest.mark.parametrize("n,theta2", [(1, 0.25), (2, 0.25), (9, 1.0), (1000, 0.25), (20000, 0.25)])
def test_barnes_hut_f64_device_tree(gpu, orc, n, theta2, leaf):
    """NBODY_TREE_DEVICE on an f64 handle (kernels_tree.hip instantiated for double): the same cells, pre-order, skip
    links, widths and leaves as the oracle's f64 tree, bit for bit; centres of mass from f64 prefix sums instead of the
    reference's sequential f64 folds (1e-13); the strict f64 walk on it visits the same nodes unless an opening test sits
    within that rounding of its threshold."""
    nb = gpu
    sd = dict(g=1.0, g_soft=0.01, dt=1e-3, theta2=theta2)
    ics = nb.plummer(n, seed=53, f64=True)
    ref = ics.copy().astype(orc.P64)
    lm = 1 if leaf == "direct" else 0
    acc_n, vis_n = orc.bh_update_forces(ref, sd, BOX[0], BOX[1], threads=4, leaf_mode=lm)
    rt = orc.bh_build_tree(ics.astype(orc.P64), BOX[0], BOX[1])
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, tree_build=nb.TREE_DEVICE, leaf_mode=nb.LEAF_DIRECT if lm else nb.LEAF_REFERENCE) as sim:
        sim.settings = nb.Settings(**sd)
        sim.update_forces()
        got = sim.get_points()
        s = sim.stats()
        t = sim.tree()
    assert s.tree_nodes == len(rt["skip"]) and np.array_equal(t["skip"], rt["skip"]) and eq(t["width"], rt["width"])
    leaves = rt["nchild"] == 0
    assert eq(t["com_mass"][leaves], rt["com_mass"][leaves])
    assert np.abs(t["com_mass"][:, :3] - rt["com_mass"][:, :3]).max(initial=0.0) < 1e-12 * BOX[1]
    # (the reference's own sequential fold over n terms carries up to n * 2^-53 of rounding: the bound on the difference)
    assert np.abs(t["com_mass"][:, 3] - rt["com_mass"][:, 3]).max(initial=0.0) < max(4e-15, n * 2.3e-16)
    assert abs(int(s.interactions) - acc_n) <= max(1, 1e-6 * acc_n) and abs(int(s.node_visits) - vis_n) <= max(1, 1e-6 * vis_n)
    scale = np.abs(ref["acceleration"]).max() or 1.0
    err = np.abs(got["acceleration"] - ref["acceleration"]).max(axis=1) / scale
    assert np.count_nonzero(err > 1e-11) <= 1 and err.max(initial=0.0) < 1e-3


def test_barnes_hut_f64_device_tree_trajectory_and_close_pairs(gpu, orc):
    """8 steps with escapes in a tight box, with a pair of bodies 1e-9 apart (they share all 21 levels of the first keys:
    the second keys order them): survivors equal the oracle's, positions to 1e-12."""
    nb = gpu
    box = ((0.0, 0.0, 0.0), 2.0)
    sd = dict(g=1.0, g_soft=0.05, dt=2e-2, theta2=0.5)
    ics = nb.plummer(3000, seed=54, f64=True)
    ics["position"][17] = ics["position"][5] + 1e-9
    ics["velocity"][17] = ics["velocity"][5]
    ref = ics.copy().astype(orc.P64)
    with nb.Simulation(ics, *box, method=nb.BARNES_HUT, tree_build=nb.TREE_DEVICE) as sim:
        sim.settings = nb.Settings(**sd)
        sim.init()
        for _ in range(8):
            sim.step()
            ref, _, _ = orc.bh_step_by(ref, sd, box[0], box[1], sd["dt"], threads=2)
            assert len(sim) == len(ref)
        got = sim.get_points()
        depth = np.log2(box[1] / sim.tree()["width"].min())
    assert len(ref) < 2800 and depth > 22
    assert np.abs(got["position"] - ref["position"]).max() < 1e-11
    assert np.array_equal(got["mass"], ref["mass"])


@pytest.mark.parametrize("tree", ["host", "device"])
@pytest.mark.parametrize("leaf", ["reference", "direct"])
@pytest.mark.parametrize("n,theta2,split", [(1, 0.25, 0), (2, 0.25, 0), (9, 1.0, 0), (1000, 0.25, 0), (5000, 0.5, 1), (20000, 0.25, 0), (65536, 0.25, 0), (65536, 0.25, 7)])
def test_barnes_hut_f64_fast_walk(gpu, orc, n, theta2, leaf, tree, split):
    """NBODY_MATH_FAST on an f64 handle: one running sum per lane (FMA, 1/sqrt) over a node range split into segments, instead
    of the reference's nested sums.  The opening tests are the reference's, so on the host-built tree the node counts equal
    the oracle's; accelerations agree to f64 rounding -- 1e-12 of the largest, and every body to 1e-9 of its OWN |a|."""
    nb = gpu
    sd = dict(g=1.0, g_soft=0.01, dt=1e-3, theta2=theta2)
    ics = nb.plummer(n, seed=58, f64=True)
    ref = ics.copy().astype(orc.P64)
    acc_n, vis_n = orc.bh_update_forces(ref, sd, BOX[0], BOX[1], threads=8, leaf_mode=1 if leaf == "direct" else 0)
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE if tree == "device" else nb.TREE_HOST,
                       leaf_mode=nb.LEAF_DIRECT if leaf == "direct" else nb.LEAF_REFERENCE, tuning=dict(bh_walk_split=split)) as sim:
        assert sim.f64
        sim.settings = nb.Settings(**sd)
        sim.update_forces()
        got = sim.get_points()
        s = sim.stats()
    if tree == "host":
        assert (s.interactions, s.node_visits) == (acc_n, vis_n)
    else:
        assert abs(s.interactions - acc_n) <= max(2, 1e-6 * acc_n) and abs(s.node_visits - vis_n) <= max(2, 1e-6 * vis_n)
    a, r = got["acceleration"], ref["acceleration"]
    own = np.maximum(np.linalg.norm(r, axis=1), 1e-300)
    err = np.linalg.norm(a - r, axis=1) / own
    if tree == "host":
        assert rel_err(a, r) < 1e-12 and err.max() < 1e-9, (rel_err(a, r), err.max())
    else:   # a centre of mass in its last bits can flip a test that sits on its threshold (as for f32, far rarer)
        assert np.median(err) < 1e-12 and np.count_nonzero(err > 1e-9) <= max(1, n // 20000), (np.median(err), err.max())


def test_barnes_hut_f64_fast_trajectory_tracks_the_strict_one(gpu):
    """20 steps in a tight box (bodies leave): the fast f64 run, device tree and host tree, stays with the bit-exact strict run."""
    nb = gpu
    box = ((0.0, 0.0, 0.0), 3.0)
    st = nb.Settings(1.0, 0.05, 1e-2, 0.25)
    ics = nb.plummer(6000, seed=59, f64=True)
    runs = {}
    for name, kw in (("strict", dict(math_mode=nb.STRICT)), ("fast host", dict(math_mode=nb.FAST, tree_build=nb.TREE_HOST)), ("fast", dict(math_mode=nb.FAST))):
        with nb.Simulation(ics, *box, method=nb.BARNES_HUT, **kw) as sim:
            sim.settings = st
            sim.init()
            sim.steps(20)
            runs[name] = sim.get_points()
    assert len(runs["strict"]) < 6000
    for name in ("fast host", "fast"):
        assert len(runs[name]) == len(runs["strict"])
        assert np.array_equal(runs[name]["mass"], runs["strict"]["mass"])
        assert np.abs(runs[name]["position"] - runs["strict"]["position"]).max() < 1e-11, name
