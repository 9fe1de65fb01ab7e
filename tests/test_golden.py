"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the
oracle; the reference itself has none -- "parity unpinned").  CPU: the oracle still reproduces
them bit for bit and f32 tracks f64.  GPU: the HIP path (strict math) reproduces the f32 vectors bit for
bit -- brute force, Barnes-Hut with the src/manual leaf rule ("bh") and with the src/llm one ("bhd"),
node counts included."""
import glob
import os

import numpy as np
import pytest

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def load(path):
    z = np.load(path)  # allow_pickle stays False
    g, e, dt, t2 = [float(x) for x in z["settings"]]
    return z, dict(g=g, g_soft=e, dt=dt, theta2=t2), tuple(float(c) for c in z["center"]), float(z["width"])


def test_fixture_set_is_complete():
    assert len(GOLDEN) == 16


@pytest.mark.parametrize("path", GOLDEN, ids=os.path.basename)
def test_oracle_reproduces_golden(orc, path):
    z, st, center, width = load(path)
    a = z["ics"].astype(orc.P32) if str(z["ftype"]) == "f32" else orc.to_f64(z["ics"])
    counts = []
    for _ in range(int(z["steps"])):
        if str(z["kind"]) == "bf":
            a = orc.bf_step_by(a, st, center, width, st["dt"])
        else:
            a, acc, vis = orc.bh_step_by(a, st, center, width, st["dt"], threads=2, leaf_mode=1 if str(z["kind"]) == "bhd" else 0)
            counts.append((acc, vis))
    for f in ("position", "velocity", "acceleration", "mass"):
        assert np.array_equal(a[f], z[f]), f
    if counts:
        assert np.array_equal(np.array(counts, dtype=np.uint64), z["counts"])


@pytest.mark.parametrize("stem", ["bf_n64", "bf_n256"])
def test_f32_golden_tracks_f64_golden(stem):
    here = os.path.dirname(GOLDEN[0])
    a = np.load(os.path.join(here, f"{stem}_f32_t050.npz"))
    b = np.load(os.path.join(here, f"{stem}_f64_t050.npz"))
    assert np.abs(a["position"] - b["position"]).max() < 5e-6
    scale = np.abs(b["acceleration"]).max()
    assert np.abs(a["acceleration"] - b["acceleration"]).max() / scale < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=os.path.basename)
def test_hip_path_reproduces_golden(gpu, orc, path):
    """f32 AND f64 fixtures: an f64 fixture runs on an NBODY_F64 handle (80-byte records, f64 settings)."""
    nb = gpu
    z, st, center, width = load(path)
    kind = str(z["kind"])
    f64 = str(z["ftype"]) == "f64"
    method = nb.BRUTE_FORCE if kind == "bf" else nb.BARNES_HUT
    ics = orc.to_f64(z["ics"]).astype(nb.PARTICLE_DTYPE64) if f64 else z["ics"]
    with nb.Simulation(ics, center, width, method=method, math_mode=nb.STRICT,
                       leaf_mode=nb.LEAF_DIRECT if kind == "bhd" else nb.LEAF_REFERENCE) as sim:
        assert sim.f64 == f64
        sim.settings = nb.Settings(**st)
        sim.init()
        sim.steps(int(z["steps"]))
        got = sim.get_points()
        s = sim.stats()
    assert len(got) == len(z["mass"])
    if kind != "bf":
        assert s.interactions == int(z["counts"][:, 0].sum()) and s.node_visits == int(z["counts"][:, 1].sum())
    # strict math reproduces the oracle's rounding sequence on every path: brute force (ascending partner
    # order), Barnes-Hut with the reference's nested sums, and the src/llm leaf rule's single running sum
    bits = np.uint64 if f64 else np.uint32
    for f in ("position", "velocity", "acceleration", "mass"):
        assert got[f].dtype == z[f].dtype
        assert np.array_equal(got[f].view(bits), z[f].view(bits)), f
