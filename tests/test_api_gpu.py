"""ABI behaviour around the hot path: strides, re-upload, capacity, error reporting, statistics."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
BOX = ((0.0, 0.0, 0.0), 64.0)


def test_strided_upload_and_download(gpu):
    """Callers may embed PointParticle in a larger record: stride >= 40, multiple of 4."""
    nb = gpu
    n = 300
    ics = nb.plummer(n, seed=1)
    wide = np.zeros(n, dtype=np.dtype([("p", nb.PARTICLE_DTYPE), ("tag", "<i4", 2)]))
    wide["p"] = ics
    wide["tag"] = 7
    with nb.Simulation(ics[:1], *BOX, capacity=n) as sim:
        sim._check(nb.lib.nbody_upload(sim._h, wide.ctypes.data, n, wide.dtype.itemsize))
        out = np.zeros(n, dtype=wide.dtype)
        out["tag"] = 9
        got_n = C.c_size_t(0)
        sim._check(nb.lib.nbody_download(sim._h, out.ctypes.data, n, out.dtype.itemsize, C.byref(got_n)))
        assert got_n.value == n
        assert np.array_equal(out["p"], ics) and np.all(out["tag"] == 9)
        assert nb.lib.nbody_upload(sim._h, wide.ctypes.data, n, 38) == nb.NBODY_ERR_INVALID
        assert nb.lib.nbody_upload(sim._h, wide.ctypes.data, n + 1, 48) == nb.NBODY_ERR_CAPACITY
        small = np.zeros(10, dtype=nb.PARTICLE_DTYPE)
        assert nb.lib.nbody_download(sim._h, small.ctypes.data, 10, 40, C.byref(got_n)) == nb.NBODY_ERR_CAPACITY
        assert got_n.value == n and b"too small" in nb.lib.nbody_last_error(sim._h)


def test_reupload_replaces_the_body_vector(gpu, orc):
    nb = gpu
    a, b = nb.plummer(500, seed=1), nb.plummer(200, seed=2)
    sd = dict(g=1.0, g_soft=0.0, dt=1e-3, theta2=0.5)
    with nb.Simulation(a, *BOX, math_mode=nb.STRICT) as sim:
        sim.steps(2)
        sim._check(nb.lib.nbody_upload(sim._h, b.ctypes.data, len(b), 40))
        assert len(sim) == 200
        sim.steps(2)
        got = sim.get_points()
    ref = b.astype(orc.P32)
    for _ in range(2):
        ref = orc.bf_step_by(ref, sd, BOX[0], BOX[1], sd["dt"])
    assert np.array_equal(got["position"], ref["position"])


def test_step_without_bounds_is_an_error(gpu):
    nb = gpu
    h = C.c_void_p()
    cfg = nb.NbodyConfig(C.sizeof(nb.NbodyConfig), nb.BRUTE_FORCE, nb.STRICT, 0, -1, 0, 1, 0, 16, 0, 0)
    assert nb.lib.nbody_create(C.byref(cfg), C.byref(h)) == 0
    try:
        assert nb.lib.nbody_step_by(h, 1e-3) == nb.NBODY_ERR_INVALID
        assert b"nbody_set_bounds" in nb.lib.nbody_last_error(h)
        g, e, dt, t2 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        assert nb.lib.nbody_get_settings(h, C.byref(g), C.byref(e), C.byref(dt), C.byref(t2)) == 0
        assert (g.value, e.value, dt.value, t2.value) == (1.0, 0.0, np.float32(1e-3), 0.5)   # shared.rs:69-78
    finally:
        nb.lib.nbody_destroy(h)


def test_device_ordinal_out_of_range(gpu):
    nb = gpu
    h = C.c_void_p()
    cfg = nb.NbodyConfig(C.sizeof(nb.NbodyConfig), 0, 0, 0, 99, 0, 1, 0, 16, 0, 0)
    assert nb.lib.nbody_create(C.byref(cfg), C.byref(h)) == nb.NBODY_ERR_INVALID
    assert b"out of range" in nb.lib.nbody_last_error(None)


def test_statistics_and_profiling(gpu):
    nb = gpu
    n = 4096
    with nb.Simulation(nb.plummer(n), *BOX, math_mode=nb.FAST) as sim:
        sim.steps(3)
        s = sim.stats()
        assert (s.steps, s.interactions, s.force_launches) == (3, 3 * n * (n - 1), 0)
        sim.set_profiling(True)
        sim.reset_stats()
        sim.steps(4)
        s = sim.stats()
        assert (s.steps, s.force_launches) == (4, 4) and s.force_kernel_ms > 0
        # the timed launch is the symmetric kernel's rotation: every pair of two DIFFERENT resident sets met symmetrically (6 of
        # the 8 sets' 8 partners at this size; the own-set and opposite-set pairs are the companion blocks' share)
        assert 0.7 * 4 * n * (n - 1) <= s.force_kernel_interactions <= 4 * n * (n - 1)
    with nb.Simulation(nb.plummer(n), *BOX, math_mode=nb.FAST, tuning=dict(sym_min_bodies=1 << 30)) as sim:   # the LDS-tiled kernel: one launch, all pairs
        sim.set_profiling(True)
        sim.steps(2)
        assert sim.stats().force_kernel_interactions == 2 * n * (n - 1)
        sim.reset_stats()
        assert sim.stats().steps == 0


def test_many_handles_share_a_device(gpu, orc):
    """The visualiser keeps a pristine clone beside the running simulation (vis.rs:43,217-220)."""
    nb = gpu
    sd = dict(g=1.0, g_soft=0.0, dt=1e-3, theta2=0.5)
    sims = [nb.Simulation(nb.plummer(128, seed=s), *BOX, math_mode=nb.STRICT) for s in range(6)]
    for k, s in enumerate(sims):
        s.steps(k + 1)
    for k, s in enumerate(sims):
        ref = nb.plummer(128, seed=k).astype(orc.P32)
        for _ in range(k + 1):
            ref = orc.bf_step_by(ref, sd, BOX[0], BOX[1], sd["dt"])
        assert np.array_equal(s.get_points()["position"], ref["position"])
        s.close()


@pytest.mark.parametrize("method", ["bf", "bh"])
def test_clone_of_a_handle_with_a_live_communicator(gpu, method):
    """`Clone` (shared.rs:80): the twin of a handle whose RCCL communicator is up (a world of one here) has the state
    but not the communicator, steps on its own, and leaves the original and its communicator usable."""
    nb = gpu
    ics = nb.plummer(3000, seed=81)
    m = nb.BRUTE_FORCE if method == "bf" else nb.BARNES_HUT
    with nb.Simulation(ics, (0, 0, 0), 64.0, method=m, math_mode=nb.FAST) as sim:
        sim.settings = nb.Settings(1.0, 0.01, 1e-3, 0.25)
        sim.comm_init(nb.comm_unique_id())
        sim.steps(3)
        twin = sim.clone()
        try:
            sim.steps(2)
            twin.steps(2)
            a, b = sim.get_points(), twin.get_points()
            assert sim.elapsed() == twin.elapsed()
            for f in ("position", "velocity", "acceleration", "mass"):
                assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), f
            sim.steps(1)                      # the original's communicator survived the clone
            assert len(sim) == len(twin) == 3000
        finally:
            twin.close()
