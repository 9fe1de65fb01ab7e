"""Barnes-Hut over SPATIAL shards with a halo exchange (NBODY_SHARD_SPATIAL; BASELINE configs[4])
on ONE GPU: G handles of this process play ranks 0..G-1; the four exchanges a step makes with RCCL are done with
device-to-device copies (nbody_debug_let_*).  Everything else is the production code: ownership by Morton-key range,
migration, the distributed device build, the spanning cells, the export pruned by the partners' bounding boxes, the walk
over the array of the nodes a rank holds (its slice and the imports, in global-index order).  Oracle = the single-shard device-tree run (and through it the CPU oracle)."""
import numpy as np
import pytest


pytestmark = pytest.mark.gpu
BOX = ((0.0, 0.0, 0.0), 64.0)
FIELDS = ("position", "velocity", "acceleration", "mass")


def make_world(nb, ics, G, box, st, prune=True, **kw):
    sims = [nb.Simulation(ics, *box, method=nb.BARNES_HUT, math_mode=nb.FAST, rank=r, world_size=G, capacity=len(ics),
                          shard_mode=nb.SHARD_SPATIAL, **kw) for r in range(G)]
    for s in sims:
        s.settings = st
        s.set_prune(prune)
        s.init()
    return sims


def assert_same_up_to_flips(got, ref, tol):
    """agreement to `tol` (of the largest component) for all but a handful of bodies, and to the Barnes-Hut truncation of
    one cell for those: a last-bit difference in a centre of mass or a position can flip one opening test"""
    n = len(ref)
    err = np.abs(np.asarray(got, np.float64) - ref).max(axis=1) / np.abs(ref).max()
    far = np.count_nonzero(err > tol)
    assert far <= max(1, n // 5000) and err.max() < 1e-4, (err.max(), far)


def close(sims):
    for s in sims:
        s.close()


def host_keys(nb, ics, box):
    """the 63-bit keys of the device build (orthant codes of 21 levels), in numpy float32"""
    p = ics["position"].astype(np.float32)
    c = np.tile(np.array(box[0], np.float32), (len(p), 1))
    hw = np.float32(box[1]) * np.float32(0.5)
    key = np.zeros(len(p), np.uint64)
    for _ in range(21):
        b = p > c
        key = (key << np.uint64(3)) | (b[:, 0].astype(np.uint64) | (b[:, 1].astype(np.uint64) << np.uint64(1)) | (b[:, 2].astype(np.uint64) << np.uint64(2)))
        hw = hw * np.float32(0.5)
        c = np.where(b, c + hw, c - hw).astype(np.float32)
    return key


@pytest.mark.parametrize("G", [2, 3, 8])
def test_upload_deals_key_ranges_and_download_ids_restore_the_vector(gpu, G):
    nb = gpu
    n = 5000
    ics = nb.plummer(n, seed=61)
    sims = make_world(nb, ics, G, BOX, nb.Settings(1.0, 0.01, 1e-3, 0.25))
    key = host_keys(nb, ics, BOX)
    seen = np.zeros(n, bool)
    tops = []
    for s in sims:
        ids = s.download_ids()
        got = s.get_points()
        assert np.array_equal(got, ics[ids]) and np.all(np.diff(ids) > 0)      # own bodies in the vector's order
        assert not seen[ids].any()
        seen[ids] = True
        if len(ids):
            tops.append((key[ids].min(), key[ids].max(), len(ids)))
    assert seen.all()
    for (lo0, hi0, c0), (lo1, hi1, c1) in zip(tops, tops[1:]):
        assert hi0 < lo1                                                        # disjoint, ascending key ranges
    assert max(c for _, _, c in tops) - min(c for _, _, c in tops) <= 2        # the G-quantiles
    rec, idx = nb.spatial_gather(sims, n)
    assert np.array_equal(idx, np.arange(n)) and np.array_equal(rec, ics)
    close(sims)


@pytest.mark.parametrize("G", [2, 4])
def test_migration_keeps_ownership_by_key_range(gpu, G):
    """g = 0: straight lines, exactly reproducible in numpy float32.  Bodies cross the key-range bounds and the box walls;
    after every step each rank holds exactly the bodies whose key lies in its range, nothing is lost or duplicated.  The
    bounds are redrawn every step at the G-quantiles of the world's keys (they apply from the next step on), so the
    ranks stay balanced while the box empties."""
    nb = gpu
    n = 4000
    rng = np.random.default_rng(5)
    ics = np.zeros(n, nb.PARTICLE_DTYPE)
    ics["position"] = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32)
    ics["velocity"] = rng.normal(0.0, 1.0, (n, 3)).astype(np.float32)
    ics["mass"] = rng.uniform(0.5, 1.5, n).astype(np.float32)
    box = ((0.0, 0.0, 0.0), 2.2)
    dt = np.float32(0.05)
    sims = make_world(nb, ics, G, box, nb.Settings(0.0, 0.0, float(dt), 0.25))
    for s in sims:
        s.set_balance(False)    # quantiles of the body count (the default weighs the bodies by their walks' visit counts)
    key0 = host_keys(nb, ics, box)
    bounds = sims[0].let_bounds()
    assert [int(b) for b in bounds[1:G]] == [int(np.sort(key0)[min(n - 1, r * n // G)]) for r in range(1, G)]   # the upload's quantiles
    x = ics["position"].copy()
    alive = np.arange(n)
    half = (ics["velocity"] * np.float32(0.5)) * dt
    lo, hi = np.float32(0.0) + np.float32(-1.1), np.float32(0.0) + np.float32(1.1)
    for step in range(6):
        nb.spatial_step(sims)
        x = x + half
        keep = np.all((x >= lo) & (x <= hi), axis=1)
        x, half, alive = x[keep], half[keep], alive[keep]
        tree_pos = x.copy()                                   # where the bodies are when the tree is built
        x = x + half
        rec, idx = nb.spatial_gather(sims, n)
        assert np.array_equal(idx, alive), step
        assert np.array_equal(rec["position"].view(np.uint32), x.view(np.uint32)), step
        tmp = np.zeros(len(tree_pos), nb.PARTICLE_DTYPE)
        tmp["position"] = tree_pos
        k = host_keys(nb, tmp, box)
        owner = np.searchsorted(bounds[1:G], k, side="right")   # this step classified with the bounds drawn in the step before
        for r, s in enumerate(sims):
            ids = s.download_ids()
            assert set(ids.tolist()) == set(alive[owner == r].tolist()), (step, r)
        bounds = sims[0].let_bounds()
        assert all(np.array_equal(s.let_bounds(), bounds) for s in sims)
        ks = np.sort(k)                                       # ... and the new ones are this step's quantiles
        assert [int(b) for b in bounds[1:G]] == [int(ks[min(len(ks) - 1, r * len(ks) // G)]) for r in range(1, G)], step
    assert len(alive) < 0.9 * n
    close(sims)


@pytest.mark.parametrize("prune", [False, True])
@pytest.mark.parametrize("leaf", ["reference", "direct"])
@pytest.mark.parametrize("G,n", [(1, 3000), (2, 3000), (3, 5000), (4, 20000), (8, 20000), (8, 300), (5, 64)])
def test_spatial_forces_equal_the_single_shard_device_tree(gpu, orc, G, n, leaf, prune):
    """One force pass: the tree the G ranks assemble is the tree of the single-shard device build.  Node counts add up
    to its counts (a centre of mass may differ in its last bit: local f64 prefix sums), accelerations to rounding; with
    and without pruning the result is THE SAME BITS (pruning only leaves out nodes nobody visits)."""
    nb = gpu
    st = nb.Settings(1.0, 0.01, 1e-3, 0.25)
    ics = nb.plummer(n, seed=62)
    lm = nb.LEAF_DIRECT if leaf == "direct" else nb.LEAF_REFERENCE
    sims = make_world(nb, ics, G, BOX, st, prune=prune, leaf_mode=lm)
    nb.spatial_step(sims, forces_only=True)
    rec, idx = nb.spatial_gather(sims, n)
    stats = [s.stats() for s in sims]
    ls = [s.let_stats() for s in sims]
    close(sims)
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE, leaf_mode=lm) as one:
        one.settings = st
        one.update_forces()
        ref = one.get_points()
        s1 = one.stats()
    assert np.array_equal(idx, np.arange(n))
    assert all(s.tree_nodes == s1.tree_nodes for s in stats)
    acc = sum(s.interactions for s in stats)
    vis = sum(s.node_visits for s in stats)
    assert abs(acc - s1.interactions) <= max(2, 2e-6 * s1.interactions) and abs(vis - s1.node_visits) <= max(2, 2e-6 * s1.node_visits)
    # rounding for all but the odd body for which a centre of mass that differs in its last bit flips one opening test
    # (w^2 < theta^2 r^2 sits on the threshold): that body's error is the Barnes-Hut truncation of one cell, not more
    assert_same_up_to_flips(rec["acceleration"], ref["acceleration"], 2e-6)
    assert np.array_equal(rec["position"], ref["position"])
    # ... and against the oracle itself (the CPU restatement of the reference walk on the reference's tree), per body: the
    # median error of a body relative to its OWN |a| is rounding (one running f32 sum over its ~2 000 accepted nodes: the walk
    # of a spatial rank is not split into node-range segments), and no body is further off than one flipped opening test
    oref = ics.copy().astype(orc.P32)
    orc.bh_update_forces(oref, dict(g=1.0, g_soft=0.01, dt=1e-3, theta2=0.25), BOX[0], BOX[1], threads=8, leaf_mode=1 if leaf == "direct" else 0)
    oa = oref["acceleration"].astype(np.float64)
    own = np.maximum(np.linalg.norm(oa, axis=1), 1e-30)
    err = np.linalg.norm(rec["acceleration"].astype(np.float64) - oa, axis=1) / own
    assert np.median(err) < 5e-6 and np.count_nonzero(err > 1e-4) <= max(4, n // 2000) and err.max() < 5e-3, (np.median(err), err.max())
    if G > 1 and not prune:   # every private node to every partner that has bodies (the <= 21 spanning cells per rank stay home)
        assert all(l.nodes_local * (G - 1) - 21 * (G - 1) <= l.nodes_sent <= l.nodes_local * (G - 1) for l in ls)


def test_spatial_shards_with_bodies_that_share_all_21_levels(gpu):
    """Groups of bodies a few 1e-7 apart (equal 63-bit keys: always on one rank) get the device build's second keys in
    the distributed build too; without them (nbody_tree_max_tie = 1) the step is refused, there being no host build to
    fall back to."""
    nb = gpu
    n, G = 6000, 3
    st = nb.Settings(1.0, 0.01, 1e-3, 0.25)
    ics = nb.plummer(n, seed=66)
    rng = np.random.default_rng(1)
    for dst, src in ((1, 0), (7, 3), (11, 3), (n - 1, n // 2), (100, 99), (101, 99), (102, 99), (103, 99)):
        ics["position"][dst] = ics["position"][src] + rng.integers(1, 12, 3).astype(np.float32) * np.float32(1.2e-7)
    sims = make_world(nb, ics, G, BOX, st)
    nb.spatial_step(sims, forces_only=True)
    rec, idx = nb.spatial_gather(sims, n)
    stats = [s.stats() for s in sims]
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE) as one:
        one.settings = st
        one.update_forces()
        ref = one.get_points()
        s1 = one.stats()
        width = one.tree()["width"]
    assert np.log2(BOX[1] / np.float64(width.min())) > 22
    assert all(s.tree_nodes == s1.tree_nodes for s in stats)
    assert_same_up_to_flips(rec["acceleration"], ref["acceleration"], 2e-6)
    for s in sims:
        s.set_tuning("tree_max_tie", 1)
    try:
        with pytest.raises(nb.NbodyError) as e:
            nb.spatial_step(sims, forces_only=True)
        assert e.value.code == nb.NBODY_ERR_TREE_DEPTH
    finally:
        close(sims)


def test_bounds_weighted_by_the_walks_visit_counts_even_out_the_work(gpu):
    """Equal body counts are not equal work: bodies in the dense core visit more nodes.  By default the bounds are redrawn
    at the quantiles of the last walk's per-body visit counts (they ride in acc.w through retain and migration)."""
    nb = gpu
    n, G = 40000, 4
    st = nb.Settings(1.0, 0.01, 1e-3, 0.25)
    ics = nb.plummer(n, seed=67)
    ics["position"] += np.float32([0.37, -0.21, 0.13])       # off the octant symmetry: the key ranges get different densities
    spread = {}
    for by_work in (False, True):
        sims = make_world(nb, ics, G, BOX, st)
        for s in sims:
            s.set_balance(by_work)
        for _ in range(4):
            nb.spatial_step(sims)
        for s in sims:
            s.reset_stats()
        nb.spatial_step(sims)
        visits = np.array([s.stats().node_visits for s in sims], np.float64)
        owned = [len(s) for s in sims]
        spread[by_work] = visits.max() / visits.mean()
        rec, idx = nb.spatial_gather(sims, n)
        assert np.array_equal(idx, np.arange(n))
        close(sims)
        print(f"balance by {'work' if by_work else 'count'}: bodies {owned}, visits max/mean {spread[by_work]:.3f}")
    assert spread[True] < 1.03 and spread[True] < spread[False]


def test_pruning_changes_the_volume_not_the_result(gpu):
    nb = gpu
    n, G = 20000, 4
    st = nb.Settings(1.0, 0.01, 1e-3, 0.25)
    ics = nb.plummer(n, seed=63)
    out = {}
    for prune in (False, True):
        sims = make_world(nb, ics, G, BOX, st, prune=prune)
        for _ in range(3):
            nb.spatial_step(sims)
        rec, idx = nb.spatial_gather(sims, n)
        out[prune] = (rec, idx, [s.stats() for s in sims], [s.let_stats() for s in sims])
        close(sims)
    a, b = out[False], out[True]
    assert np.array_equal(a[1], b[1])
    for f in FIELDS:
        assert np.array_equal(a[0][f].view(np.uint32), b[0][f].view(np.uint32)), f
    assert [(s.interactions, s.node_visits) for s in a[2]] == [(s.interactions, s.node_visits) for s in b[2]]
    sent_all = sum(l.nodes_sent for l in a[3])
    sent_pruned = sum(l.nodes_sent for l in b[3])
    assert 0 < sent_pruned < 0.6 * sent_all
    print(f"nodes exported over 3 steps, 4 ranks: unpruned {sent_all}, pruned {sent_pruned} ({sent_pruned / sent_all:.1%})")


def test_node_list_buffers_grow_when_a_step_needs_more(gpu):
    """The export lists (one after the other in one buffer) and the staged imports start at a quarter of the slice's node
    capacity and grow on demand -- the export lists are then written again from the per-node partner masks.  With ~1 000
    records to start with every step grows something: the bits do not change, and with pruning off (every node to every
    partner: the largest lists there are) neither."""
    nb = gpu
    n, G = 20000, 4
    st = nb.Settings(1.0, 0.01, 1e-3, 0.25)
    ics = nb.plummer(n, seed=63)
    for prune in (True, False):
        out = []
        for tuning in (None, {"let_list_div": 1 << 20}):
            sims = make_world(nb, ics, G, BOX, st, prune=prune, tuning=tuning)
            for _ in range(3):
                nb.spatial_step(sims)
            rec, idx = nb.spatial_gather(sims, n)
            out.append((rec, idx, [s.let_stats() for s in sims]))
            close(sims)
        a, b = out
        assert np.array_equal(a[1], b[1])
        for f in FIELDS:
            assert np.array_equal(a[0][f].view(np.uint32), b[0][f].view(np.uint32)), (prune, f)
        assert [l.nodes_received for l in a[2]] == [l.nodes_received for l in b[2]]
        assert min(l.nodes_received for l in b[2]) > 3 * 2048    # (more per step than the small buffers held at first)


@pytest.mark.parametrize("G,n,box_w", [(2, 3000, 64.0), (4, 20000, 64.0), (3, 9000, 2.5), (8, 20000, 3.0)])
def test_spatial_trajectory_tracks_the_single_shard_run(gpu, orc, G, n, box_w):
    """Several steps, with bodies leaving a tight box and migrating between ranks: the world stays with the single-shard
    device-tree run (same survivors; positions to rounding) and, through it, with the oracle."""
    nb = gpu
    box = ((0.0, 0.0, 0.0), box_w)
    sd = dict(g=1.0, g_soft=0.01, dt=5e-3, theta2=0.25)
    st = nb.Settings(**sd)
    ics = nb.plummer(n, seed=64)
    sims = make_world(nb, ics, G, box, st)
    with nb.Simulation(ics, *box, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE) as one:
        one.settings = st
        one.init()
        for _ in range(6):
            nb.spatial_step(sims)
            one.step()
        ref = one.get_points()
        s1 = one.stats()
    rec, idx = nb.spatial_gather(sims, n)
    stats = [s.stats() for s in sims]
    ls = [s.let_stats() for s in sims]
    assert sum(len(s) for s in sims) == sims[0].count_global() == len(ref) == len(rec)
    if box_w < 10:
        assert len(ref) < n
    assert np.array_equal(rec["mass"], ref["mass"])
    assert np.abs(rec["position"].astype(np.float64) - ref["position"]).max() < 2e-6
    assert_same_up_to_flips(rec["acceleration"], ref["acceleration"], 1e-5)
    acc = sum(s.interactions for s in stats)
    assert abs(acc - s1.interactions) <= 1e-5 * s1.interactions
    assert all(abs(s.elapsed() - 6 * sd["dt"]) < 1e-6 for s in sims)
    ref_o = ics.copy().astype(orc.P32)
    for _ in range(6):
        ref_o, _, _ = orc.bh_step_by(ref_o, sd, box[0], box[1], sd["dt"], threads=4)
    assert len(ref_o) == len(rec) and np.abs(rec["position"].astype(np.float64) - ref_o["position"]).max() < 1e-5
    close(sims)


def test_spatial_handles_refuse_what_they_do_not_support(gpu):
    nb = gpu
    ics = nb.plummer(64)
    for kw in (dict(method=nb.BRUTE_FORCE, math_mode=nb.FAST), dict(method=nb.BARNES_HUT, math_mode=nb.STRICT)):
        with pytest.raises(nb.NbodyError) as e:
            nb.Simulation(ics, *BOX, rank=0, world_size=2, capacity=64, shard_mode=nb.SHARD_SPATIAL, **kw)
        assert e.value.code == nb.NBODY_ERR_INVALID
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.FAST, rank=0, world_size=2, capacity=64,
                       shard_mode=nb.SHARD_SPATIAL) as sim:
        with pytest.raises(nb.NbodyError) as e:
            sim.step()                      # no communicator
        assert e.value.code == nb.NBODY_ERR_COMM
        with pytest.raises(nb.NbodyError) as e:
            sim.add_point(ics[0])           # collective: needs the communicator too
        assert e.value.code == nb.NBODY_ERR_COMM
        twin = sim.clone()                  # Clone is a supertrait of Simulation (shared.rs:80): bodies, indices, bounds
        assert np.array_equal(twin.get_points(), sim.get_points()) and np.array_equal(twin.download_ids(), sim.download_ids())
        assert np.array_equal(twin.let_bounds(), sim.let_bounds())
        twin.close()
        for call in (sim.tree, sim.energy):   # a spatial rank holds neither the whole tree nor all bodies: no silent partial answers
            with pytest.raises(nb.NbodyError) as e:
                call()
            assert e.value.code == nb.NBODY_ERR_INVALID


def test_spatial_single_rank_with_a_communicator(gpu):
    """A world of one WITH a communicator runs every collective of the step (the exchanges are guarded by "has a
    communicator", not by G > 1): ncclCommInitRank, the in-place all-gathers of counts and tables, empty send/recv
    groups; results = the plain device-tree run."""
    nb = gpu
    st = nb.Settings(1.0, 0.01, 1e-3, 0.25)
    ics = nb.plummer(3000, seed=65)
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.FAST, shard_mode=nb.SHARD_SPATIAL) as sim:
        sim.settings = st
        sim.comm_init(nb.comm_unique_id())
        sim.steps(3)
        got = sim.get_points()
        ids = sim.download_ids()
    with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE) as one:
        one.settings = st
        one.steps(3)
        ref = one.get_points()
    assert np.array_equal(ids, np.arange(3000))
    assert np.abs(got["position"].astype(np.float64) - ref["position"]).max() < 1e-6


def test_randomised_worlds_against_the_single_gpu_run():
    """tools/let_stress.py: 150 random worlds (1-8 ranks, 1-30 000 bodies, spheres / clumps / lines / planes / one octant /
    near pairs, four box sizes, four thetas, both leaf rules, up to 6 steps with escapes and migration), each compared
    step by step with the single-GPU device-tree run until their trajectories part by rounding."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "let_stress.py"), "--cases", "150", "--seed", "3"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all OK" in r.stdout


def test_spatial_ranks_with_a_clump_inside_one_level_16_cell(gpu):
    """More than 256 bodies that share 16 levels of the tree (kernels_tree.hip sorts such a group with a workgroup): the spatial
    world still equals the one-handle device-tree run; beyond 4096 the rank that holds the clump says what it is."""
    nb = gpu
    st = nb.Settings(1.0, 1e-3, 1e-3, 0.25)
    rng = np.random.default_rng(7)
    for clump, ok in ((1500, True), (9000, False)):   # (9000: whatever the cut, one of the three ranks holds more than 4096 of them)
        ics = nb.plummer(6000 + clump, seed=92)
        w16 = np.float64(BOX[1]) / 65536.0
        ics["position"][6000:] = (-32.0 + 36000 * w16 + w16 * (0.1 + 0.5 * rng.random((clump, 3)))).astype(np.float32)
        ics["velocity"][6000:] = 0.0
        sims = make_world(nb, ics, 3, BOX, st)
        if ok:
            nb.spatial_step(sims, forces_only=True)
            rec, idx = nb.spatial_gather(sims, len(ics))
            with nb.Simulation(ics, *BOX, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE) as one:
                one.settings = st
                one.update_forces()
                ref = one.get_points()
                s1 = one.stats()
            assert all(s.stats().tree_nodes == s1.tree_nodes for s in sims)
            assert np.abs(rec["acceleration"].astype(np.float64) - ref["acceleration"]).max() < 1e-3 * np.abs(ref["acceleration"]).max()
        else:
            with pytest.raises(nb.NbodyError) as e:
                nb.spatial_step(sims, forces_only=True)
            assert e.value.code == nb.NBODY_ERR_TREE_DEPTH and "4096 bodies" in str(e.value)
        close(sims)
