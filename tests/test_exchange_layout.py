"""Host arithmetic of the spatial step's variable-size rounds (nbody_let.cpp exchange_layout, through the C ABI's
nbody_host_exchange_layout): from the all-gathered G x G count matrix every rank derives where each of its messages
starts and how long it is.  A send and the receive that meets it must carry the same count (RCCL hangs or corrupts
otherwise), clamped alike on both sides; a rank's messages must tile its buffers without overlap, in rank order
(the order k_let_pack_migrants packs the emigrants: d_send_off is the running sum of the send counts)."""
import numpy as np
import pytest


@pytest.mark.parametrize("G", [1, 2, 3, 4, 5, 8, 16])
@pytest.mark.parametrize("packed", [True, False])
def test_every_send_meets_a_receive_of_the_same_size(nb, G, packed):
    rng = np.random.default_rng(100 * G + packed)
    for trial in range(20):
        m = rng.integers(0, 5000, (G, G)).astype(np.int32)
        m[rng.random((G, G)) < 0.3] = 0
        if trial % 3 == 0:
            m[rng.integers(0, G), rng.integers(0, G)] = 10 ** 6      # over the clamp
        np.fill_diagonal(m, rng.integers(0, 7, G))                    # (what a rank "sends itself" must be ignored)
        clamp, stride = 4096, 5000
        lay = [nb.host_exchange_layout(m, r, clamp, packed, stride) for r in range(G)]
        for r in range(G):
            L = lay[r]
            assert L["n_out"][r] == 0 and L["n_in"][r] == 0
            for q in range(G):
                if q == r:
                    continue
                assert L["n_out"][q] == lay[q]["n_in"][r] == min(int(m[r, q]), clamp)      # the pair agrees
            # the receive buffer is tiled in rank order
            assert list(L["in_at"]) == list(np.concatenate([[0], np.cumsum(L["n_in"])[:-1]]))
            assert L["total_in"] == int(L["n_in"].sum())
            if packed:   # the send buffer is the packed run of the emigrants, destination by destination
                assert list(L["out_at"]) == list(np.concatenate([[0], np.cumsum(L["n_out"])[:-1]]))
            else:        # one list of `stride` records per partner
                assert list(L["out_at"]) == [q * stride for q in range(G)]
                assert all(L["n_out"] <= stride)


def test_bad_arguments_are_refused(nb):
    m = np.zeros((2, 2), np.int32)
    with pytest.raises(nb.NbodyError):
        nb.host_exchange_layout(m, 2, 10, True)
    with pytest.raises(nb.NbodyError):
        nb.host_exchange_layout(np.zeros((17, 17), np.int32), 0, 10, True)
