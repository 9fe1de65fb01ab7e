"""Barnes-Hut steps enqueued without any read-back (device tree, single shard) and the parallel retain (K4)."""
import os

import numpy as np
import pytest

from conftest import Knob, rel_err

pytestmark = pytest.mark.gpu
BOX = ((0.0, 0.0, 0.0), 64.0)
FIELDS = ("position", "velocity", "acceleration", "mass")


def run(nb, ics, box, st, math_mode, steps, async_on, chunks=1, leaf=None):
    old = os.environ.get("NBODY_BH_ASYNC")
    os.environ["NBODY_BH_ASYNC"] = "1" if async_on else "0"   # read by nbody_create
    try:
        kw = dict(method=nb.BARNES_HUT, math_mode=math_mode, tree_build=nb.TREE_DEVICE)
        if leaf is not None:
            kw["leaf_mode"] = leaf
        with nb.Simulation(ics, *box, **kw) as sim:
            sim.settings = st
            sim.init()
            lens = []
            for _ in range(chunks):
                sim.steps(steps // chunks)
                lens.append(len(sim))
            return sim.get_points(), sim.stats(), sim.elapsed(), lens
    finally:
        if old is None:
            os.environ.pop("NBODY_BH_ASYNC", None)
        else:
            os.environ["NBODY_BH_ASYNC"] = old


@pytest.mark.parametrize("math", ["fast", "strict"])
@pytest.mark.parametrize("n,box_w,chunks", [(3000, 2.0, 1), (3000, 2.0, 5), (20000, 64.0, 2), (65, 1.0, 3)])
def test_unsynchronised_steps_equal_synchronised_steps(gpu, math, n, box_w, chunks):
    """nbody_steps(k) on a device-tree handle enqueues k steps with no host round trip: node count, live body count
    and build flags stay on the device.  Same kernels, same order: state, counters and elapsed equal the run that
    reads the tree info back every step, bit for bit -- with bodies leaving the box on the way (the walk takes its
    body count from the device) and with the host looking at the count between chunks of steps."""
    nb = gpu
    box = ((0.0, 0.0, 0.0), box_w)
    st = nb.Settings(1.0, 0.05, 2e-2 if box_w < 10 else 1e-3, 0.25)
    ics = nb.plummer(n, seed=41)
    mm = nb.FAST if math == "fast" else nb.STRICT
    # fast math: the automatic number of walk segments follows the node count, which the unsynchronised path only
    # bounds from above; another split groups the partial sums differently, so it is pinned here
    spl = Knob(nb, "bh_walk_split", 0)
    spl.value = 4 if n < 1000 else 16
    try:
        a, sa, ea, la = run(nb, ics, box, st, mm, 30, True, chunks)
        b, sb, eb, lb = run(nb, ics, box, st, mm, 30, False, chunks)
    finally:
        spl.value = 0
    assert la == lb and ea == eb
    if box_w < 10:
        assert len(a) < n
    for f in FIELDS:
        assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), f
    assert (sa.steps, sa.interactions, sa.node_visits, sa.tree_nodes) == (sb.steps, sb.interactions, sb.node_visits, sb.tree_nodes)


def test_poisoned_run_is_replayed_from_the_failed_step(gpu, orc):
    """Two bodies 2e-7 apart with the same velocity separate only below the 21 levels of the device build's first keys (its second keys are switched off here), step after step:
    the build raises its flag ON THE DEVICE, every later kernel of the 6 enqueued steps does nothing, and at the next
    synchronisation point the host finishes the failed step with the host build and enqueues the rest again -- which
    fails again, and so on.  Every step thus runs on the host-built tree: the run equals the strict oracle's, bit for
    bit, and the run that reads the flag back every step."""
    nb = gpu
    st = nb.Settings(1.0, 0.01, 1e-3, 0.25)
    sd = dict(g=1.0, g_soft=0.01, dt=1e-3, theta2=0.25)
    ics = nb.plummer(500, seed=3)
    ics["position"][7] = ics["position"][3] + np.float32(2e-7)
    ics["velocity"][7] = ics["velocity"][3]
    tie = Knob(nb, "tree_max_tie", 64)
    tie.value = 1   # (no second keys: any collision of the 63-bit keys is "too deep", as a group of > 64 would be)
    try:
        a, sa, ea, _ = run(nb, ics, BOX, st, nb.STRICT, 6, True)
        b, sb, eb, _ = run(nb, ics, BOX, st, nb.STRICT, 6, False)
    finally:
        tie.value = 64
    assert ea == eb and sa.steps == sb.steps == 6
    ref = ics.copy().astype(orc.P32)
    tot_a = tot_v = 0
    for _ in range(6):
        ref, acc_n, vis_n = orc.bh_step_by(ref, sd, BOX[0], BOX[1], sd["dt"], threads=2)
        tot_a += acc_n
        tot_v += vis_n
    assert np.abs(ref["position"][7].astype(np.float64) - ref["position"][3]).max() < 1e-6   # still that close at the end
    for f in FIELDS:
        assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), f
        assert np.array_equal(a[f].view(np.uint32), ref[f].view(np.uint32)), f
    assert (sa.interactions, sa.node_visits) == (sb.interactions, sb.node_visits) == (tot_a, tot_v)


def test_poison_in_the_middle_of_a_run(gpu):
    """The same, but the pair only gets that close at step 3 of 10 (built from the velocities): steps 0-2 are
    confirmed, step 3 is finished on the host, 4-9 are enqueued again."""
    nb = gpu
    st = nb.Settings(0.0, 0.0, 1e-2, 0.25)            # g = 0: straight lines, so the encounter can be placed exactly
    ics = nb.plummer(400, seed=5)
    ics["position"] *= 0.25
    ics["velocity"] *= 0.01
    # body 9 meets body 4 after 3.5 steps (the tree is built after the half drift of step 3)
    ics["velocity"][4] = 0.0
    target = ics["position"][4] + np.array([1e-6, 0.0, 0.0], np.float32)   # (never exactly coincident: that is an error)
    ics["velocity"][9] = (target - ics["position"][9]) / np.float32(3.5 * 1e-2)
    spl = Knob(nb, "bh_walk_split", 0)
    spl.value = 4
    tie = Knob(nb, "tree_max_tie", 64)
    tie.value = 1   # (no second keys: the collision poisons the run)
    try:
        a, sa, ea, _ = run(nb, ics, BOX, st, nb.FAST, 10, True)
        b, sb, eb, _ = run(nb, ics, BOX, st, nb.FAST, 10, False)
    finally:
        spl.value = 0
        tie.value = 64
    # the encounter did happen below level 21: replay the straight lines in float32
    x4, x9 = ics["position"][4].copy(), ics["position"][9].copy()
    h9 = (ics["velocity"][9] * np.float32(0.5)) * np.float32(1e-2)
    for _ in range(7):
        x9 = x9 + h9
    assert 0 < np.abs(x9.astype(np.float64) - x4).max() < 64.0 / 2 ** 22
    assert ea == eb and sa.steps == sb.steps == 10
    for f in FIELDS:
        assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), f
    assert sa.node_visits == sb.node_visits


@pytest.mark.parametrize("n,method", [(1 << 21, "BARNES_HUT"), (300000, "BRUTE_FORCE"), (1000, "BRUTE_FORCE"), (1025, "BARNES_HUT")])
def test_parallel_retain_keeps_order_at_large_n(gpu, n, method):
    """K4 as one pass over many workgroups (decoupled look-back, in place): with g = 0 a step is x += (v/2) dt twice
    with the retain in between, which numpy reproduces exactly in float32.  ~10 % of the bodies leave in one step."""
    nb = gpu
    rng = np.random.default_rng(n)
    ics = np.zeros(n, nb.PARTICLE_DTYPE)
    ics["position"] = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32)
    ics["velocity"] = rng.normal(0.0, 1.0, (n, 3)).astype(np.float32)
    ics["mass"] = rng.uniform(0.5, 1.5, n).astype(np.float32)
    dt = np.float32(0.1)
    box = ((0.0, 0.0, 0.0), 2.1)
    kw = dict(method=getattr(nb, method), math_mode=nb.FAST)
    if method == "BARNES_HUT":
        kw["tree_build"] = nb.TREE_DEVICE
    with nb.Simulation(ics, *box, **kw) as sim:
        sim.settings = nb.Settings(0.0, 0.0, float(dt), 0.25)
        sim.init()
        sim.steps(2)
        got = sim.get_points()
        assert len(sim) == len(got)
    x = ics["position"].copy()
    v = ics["velocity"]
    m = ics["mass"]
    half = (v * np.float32(0.5)) * dt
    lo, hi = np.float32(0.0) + np.float32(-1.05), np.float32(0.0) + np.float32(1.05)
    for _ in range(2):
        x = x + half
        keep = np.all((x >= lo) & (x <= hi), axis=1)
        x, half, m = x[keep], half[keep], m[keep]
        x = x + half                                   # (a = 0: the kick leaves v alone)
    assert 0.5 * n < len(x) < 0.95 * n and len(got) == len(x)
    assert np.array_equal(got["mass"], m)
    assert np.array_equal(got["position"].view(np.uint32), x.view(np.uint32))


def test_interactions_follow_the_live_counts_after_escapes(gpu, orc):
    """NbodyStats::interactions of a brute-force run: n (n - 1) per force pass with the LIVE n (the host's view of the
    count is an upper bound while bodies leave the box inside nbody_steps)."""
    nb = gpu
    box = ((0.0, 0.0, 0.0), 1.5)
    sd = dict(g=1.0, g_soft=0.05, dt=2e-2, theta2=0.5)
    ics = nb.plummer(1500, seed=11)
    ref = ics.copy().astype(orc.P32)
    want = 0
    for _ in range(10):
        ref = orc.bf_step_by(ref, sd, box[0], box[1], sd["dt"])
        want += len(ref) * (len(ref) - 1)
    for mm in (nb.STRICT, nb.FAST):
        with nb.Simulation(ics, *box, method=nb.BRUTE_FORCE, math_mode=mm) as sim:
            sim.settings = nb.Settings(**sd)
            sim.init()
            sim.steps(10)
            s = sim.stats()
            assert len(sim) == len(ref) < 1400
        assert s.interactions == want
