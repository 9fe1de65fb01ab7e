"""Host logic of the Barnes-Hut path (no GPU): the product's octree build
(nbody-llm_amd/csrc/octree_host.cpp, reached through nbody_host_build_tree) must produce the
oracle's tree (barnes_hut.rs:143-183 restated) bit for bit -- same cells, same pre-order, same
centre-of-mass rounding."""
import numpy as np
import pytest


def pos4(a):
    return np.concatenate([a["position"].astype(np.float32), a["mass"].astype(np.float32)[:, None]], axis=1)


def same_tree(nb, orc, a, center, width, threads):
    t = nb.host_build_tree(pos4(a), center, width, threads)
    r = orc.bh_build_tree(a.astype(orc.P32), center, width)
    assert len(t["width"]) == len(r["width"])
    assert np.array_equal(t["com_mass"].view(np.uint32), r["com_mass"].view(np.uint32)), "com/mass bits differ"
    assert np.array_equal(t["width"], r["width"])
    assert np.array_equal(t["skip"], r["skip"])
    assert np.array_equal(t["leaf_body"], r["leaf_body"])
    # order = body ids of the leaves in pre-order
    assert np.array_equal(t["order"], r["leaf_body"][r["leaf_body"] >= 0])
    return t


@pytest.mark.parametrize("n", [0, 1, 2, 3, 9, 64, 1000, 5000, 8192, 20000])
@pytest.mark.parametrize("threads", [1, 4])
def test_tree_matches_oracle_plummer(nb, orc, n, threads):
    a = nb.plummer(n, seed=100 + n) if n else np.zeros(0, nb.PARTICLE_DTYPE)
    same_tree(nb, orc, a, (0, 0, 0), 64.0, threads)


def test_tree_matches_oracle_disc_offcentre_box(nb, orc):
    """The reference's own workload (src/main.rs:52-101): disc in a width-10 box; plus a box whose
    centre/width are not powers of two, so centre arithmetic rounds."""
    a = nb.disc(3000, seed=5)
    same_tree(nb, orc, a, (0, 0, 0), 10.0, 3)
    same_tree(nb, orc, a, (0.1, -0.3, 0.7), 13.7, 2)


def test_tree_unequal_masses_and_clusters(nb, orc):
    rng = np.random.default_rng(3)
    n = 4000
    a = np.zeros(n, nb.PARTICLE_DTYPE)
    c = rng.normal(size=(8, 3)) * 3
    a["position"] = (c[rng.integers(0, 8, n)] + rng.normal(size=(n, 3)) * 0.01).astype(np.float32)
    a["mass"] = rng.lognormal(size=n).astype(np.float32)
    same_tree(nb, orc, a, (0, 0, 0), 64.0, 4)


def test_tree_hand_cases(nb, orc):
    a = np.zeros(2, nb.PARTICLE_DTYPE)
    a["position"] = [[0.1, 0.1, 0.1], [0.11, 0.1, 0.1]]
    a["mass"] = [1, 3]
    t = same_tree(nb, orc, a, (0, 0, 0), 2.0, 1)
    assert list(t["width"]) == [2, 1, .5, .25, .125, .0625, .03125, .015625, .015625]
    assert list(t["skip"]) == [9] * 7 + [8, 9]


def test_tree_body_outside_the_box(nb, orc):
    """update_forces() may be called on bodies that were never retained: get_orthant has no
    bounds check (shared.rs:245-254), so a lone outside body just falls into edge orthants until
    it is alone.  (Two bodies outside beyond the same corner can never be separated: the reference
    overflows its stack there; product and oracle both report the depth guard.)"""
    a = nb.plummer(200, seed=9)
    a["position"][17] = [100.0, 0.5, -0.25]
    same_tree(nb, orc, a, (0, 0, 0), 64.0, 2)
    a["position"][18] = [120.0, 0.5, -0.25]  # same y,z: only x could split them, and it never does
    with pytest.raises(nb.NbodyError) as e:
        nb.host_build_tree(pos4(a), (0, 0, 0), 64.0, 2)
    assert e.value.code == nb.NBODY_ERR_TREE_DEPTH
    with pytest.raises(RuntimeError):
        orc.bh_build_tree(a.astype(orc.P32), (0, 0, 0), 64.0)


def test_coincident_bodies_hit_the_depth_guard(nb, orc):
    """The reference recurses without bound on coincident bodies (stack overflow); both the
    product and the oracle report an error instead (documented divergence)."""
    a = np.zeros(3, nb.PARTICLE_DTYPE)
    a["position"] = [[0.3, 0.3, 0.3], [0.3, 0.3, 0.3], [1, 1, 1]]
    a["mass"] = 1
    with pytest.raises(nb.NbodyError) as e:
        nb.host_build_tree(pos4(a), (0, 0, 0), 4.0, 1)
    assert e.value.code == nb.NBODY_ERR_TREE_DEPTH
    with pytest.raises(RuntimeError):
        orc.bh_build_tree(a.astype(orc.P32), (0, 0, 0), 4.0)
