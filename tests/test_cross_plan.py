"""Host logic of the symmetric scheme across shards (no GPU): over all ranks, every pair of bodies
that live on different shards is claimed by exactly one GPU, and the send/receive lists mirror
each other (a rank that is resident for a partner sends it one message; the partner expects it)."""
import numpy as np
import pytest


@pytest.mark.parametrize("world", [2, 3, 4, 5, 6, 7, 8, 9, 14])
@pytest.mark.parametrize("seg_cap", [300, 2048, 5000, 8192])
def test_every_cross_shard_pair_is_claimed_once(nb, world, seg_cap):
    rng = np.random.default_rng(world * 1000 + seg_cap)
    n_own = [int(rng.integers(seg_cap // 3, seg_cap + 1)) for _ in range(world)]
    n_own[0] = seg_cap
    plans = [nb.host_cross_plan(r, world, seg_cap, n_own[r]) for r in range(world)]
    ipt = plans[0]["ipt"]
    assert all(p["ipt"] == ipt for p in plans), "ranks must agree on the set size"
    for r in range(world):
        for q in range(r + 1, world):
            # coverage of the block B_r x B_q at (set of 64*ipt bodies) x (chunk of 64 bodies) granularity,
            # from both sides
            claim = np.zeros((n_own[r], n_own[q]), np.int8)
            for (me, other) in ((r, q), (q, r)):
                for seg, c0, c1, a0, a1 in plans[me]["parts"]:
                    if seg != other:
                        continue
                    i0, i1 = min(n_own[me], a0 * 64 * ipt), min(n_own[me], a1 * 64 * ipt)
                    j0, j1 = min(n_own[other], c0 * 64), min(n_own[other], c1 * 64)
                    if me == r:
                        claim[i0:i1, j0:j1] += 1
                    else:
                        claim[j0:j1, i0:i1] += 1
            assert claim.min() == 1 and claim.max() == 1, (r, q)


@pytest.mark.parametrize("world", [2, 3, 4, 5, 8, 11])
def test_send_and_receive_lists_mirror_each_other(nb, world):
    seg_cap = 4096
    plans = [nb.host_cross_plan(r, world, seg_cap, seg_cap if r % 2 == 0 else 1000) for r in range(world)]
    for r, p in enumerate(plans):
        sends = [int(row[0]) for row in p["parts"]]
        assert len(set(sends)) == len(sends) and r not in sends
        for q in sends:
            assert list(plans[q]["recv_from"]).count(r) == 1
        for q in p["recv_from"]:
            assert [int(row[0]) for row in plans[int(q)]["parts"]].count(r) == 1
        # one message per partner in each direction, also when the range that partner gets is empty
        assert len(sends) == len(p["recv_from"]) == (world - 1) // 2 + (1 if world % 2 == 0 else 0)


def test_too_many_shards_is_refused(nb):
    with pytest.raises(nb.NbodyError):
        nb.host_cross_plan(0, 15, 1024, 1024)
