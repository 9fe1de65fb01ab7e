"""The launch shapes the library derives from a body count (nbody_host_launch_plan; no device needed): the fast
Barnes-Hut walk's bodies per lane x node-range segments (kernels.h walk_plan, from profiles/r03_bh_walk_plan_sweep.txt)
and the symmetric brute-force kernel's resident-set size (profiles/r03_sym_plan_sweep.txt, part 3)."""
import pytest


def test_walk_plan_follows_the_sweep(nb):
    plan = lambda n, t2=0.25, fast=True: nb.launch_plan(n, t2, fast)
    # Plummer-like walks at theta = 0.5: one body per lane while there are few, then 2, 3, 4, 6 (never back down)
    got = [plan(n)["walk_bodies_per_lane"] for n in (1000, 16384, 32768, 65536, 131072, 262144, 524288, 1 << 20, 1 << 22)]
    assert got == [1, 1, 2, 2, 2, 3, 4, 6, 6]
    assert got == sorted(got)
    # segments: ~16 384 waves' worth at one body per lane, ~32 768 lane groups' worth beyond, at most 64, at least 1
    assert plan(65536)["walk_segments"] == 64 and plan(131072)["walk_segments"] == 32 and plan(1 << 22)["walk_segments"] == 3
    for n in (1, 100, 5000, 70000, 3_000_000, 50_000_000):
        assert 1 <= plan(n)["walk_segments"] <= 64
    # a walk's length goes like theta^-3: the reference driver's disc (theta2 = 1) shares nothing at 100 000 bodies, does at 10^6
    assert plan(100_000, 1.0)["walk_bodies_per_lane"] == 1
    assert plan(1_000_000, 1.0)["walk_bodies_per_lane"] >= 2
    assert plan(20_000, 0.04)["walk_bodies_per_lane"] >= 2      # (a small opening angle: long walks)
    # strict math is the parity path: one body per lane
    assert plan(1 << 20, 0.25, fast=False)["walk_bodies_per_lane"] == 1


def test_symmetric_kernel_sets_shrink_for_small_shards(nb):
    assert [nb.launch_plan(n)["sym_bodies_per_lane"] for n in (1024, 8192, 10240, 10241, 16384, 65536)] == [4, 4, 4, 8, 8, 8]
