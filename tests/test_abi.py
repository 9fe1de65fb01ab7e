"""The C-ABI library loads without a GPU and exports every symbol include/nbody_hip.h declares;
the host-side pieces (IC generators, argument checks) behave.  No device compute here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nbody_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nbody_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree(nb):
    assert declared_symbols() == sorted(nb.DECLARED_SYMBOLS)


def test_library_exports_every_declared_symbol(nb):
    lib = ctypes.CDLL(nb.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"libnbody_hip.so lacks {missing}"
    assert lib.nbody_abi_version() == 4


def test_struct_layouts_match_the_header(nb):
    # uint32 + 7 x int32 + uint64 + 4 x int32 (48 bytes up to ABI 2, then shard_mode + reserved); 6 x uint64 + 4 x double;
    # NbodyLetStats: 8 x uint64 + 5 x double + 4 x uint64 (ABI 4)
    assert ctypes.sizeof(nb.NbodyConfig) == 56
    assert ctypes.sizeof(nb.NbodyLetStats) == 136
    assert nb.PARTICLE_DTYPE64.itemsize == 80
    assert ctypes.sizeof(nb.NbodyStats) == 80
    assert nb.PARTICLE_DTYPE.itemsize == 40
    assert [nb.PARTICLE_DTYPE.fields[k][1] for k in ("position", "velocity", "acceleration", "mass")] == [0, 12, 24, 36]


def test_library_exports_no_mutable_globals(nb):
    """Launch-shape knobs live in the handle (nbody_set_tuning); the shared object exports functions only -- no data symbol a
    second handle, a rank thread or a test could change under a running one."""
    import subprocess
    for path in (nb.LIB_PATH, nb.LIB_PATH.replace("libnbody_hip.so", "libnbody_hip_tuning.so")):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        # (what remains in the data sections under the nbody namespace are hipcc's kernel handles, `k_*`: not variables of ours)
        data = [line for line in out.splitlines()
                if re.search(r" [BDGSbdgs] nbody_", line) or (re.search(r" [BDGSbdgs] _ZN5nbody", line) and not re.search(r"_ZN5nbody\d+k_", line))]
        assert not data, data


def test_no_device_means_loud_failure(nb):
    """No CPU fallback: without a HIP device, creating a simulation fails with NBODY_ERR_NO_DEVICE."""
    if nb.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(nb.NbodyError) as e:
        nb.Simulation(nb.plummer(16), (0, 0, 0), 64.0)
    assert e.value.code == nb.NBODY_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_create_rejects_bad_configs(nb):
    h = ctypes.c_void_p()
    cfg = nb.NbodyConfig(ctypes.sizeof(nb.NbodyConfig), 7, 0, 0, -1, 0, 1, 0, 16, 0, 0)
    assert nb.lib.nbody_create(ctypes.byref(cfg), ctypes.byref(h)) == nb.NBODY_ERR_INVALID
    cfg = nb.NbodyConfig(3, 0, 0, 0, -1, 0, 1, 0, 16, 0, 0)
    assert nb.lib.nbody_create(ctypes.byref(cfg), ctypes.byref(h)) == nb.NBODY_ERR_INVALID
    cfg = nb.NbodyConfig(ctypes.sizeof(nb.NbodyConfig), 0, 0, 0, -1, 2, 2, 0, 16, 0, 0)
    assert nb.lib.nbody_create(ctypes.byref(cfg), ctypes.byref(h)) == nb.NBODY_ERR_INVALID
    assert nb.lib.nbody_step_by(None, 0.1) == nb.NBODY_ERR_INVALID


def test_plummer_ics(nb, orc):
    """G = M = 1 Henon units: total mass 1, centre of mass at rest at the origin, E ~ -1/4,
    virial ratio ~ 1; deterministic in the seed."""
    n = 4096
    a = nb.plummer(n, seed=42)
    b = nb.plummer(n, seed=42)
    c = nb.plummer(n, seed=43)
    assert np.array_equal(a, b) and not np.array_equal(a["position"], c["position"])
    assert a["mass"].sum() == pytest.approx(1.0, rel=1e-6)
    assert np.abs((a["position"] * a["mass"][:, None]).sum(0)).max() < 1e-6
    assert np.abs((a["velocity"] * a["mass"][:, None]).sum(0)).max() < 1e-6
    assert not a["acceleration"].any()
    assert np.linalg.norm(a["position"], axis=1).max() <= 10.0 + 0.1
    ke, pe = orc.energy(a.astype(orc.P32), 1.0, 0.0)
    assert ke + pe == pytest.approx(-0.25, abs=0.02)
    assert 2 * ke / -pe == pytest.approx(1.0, abs=0.06)


def test_disc_ics(nb):
    """src/main.rs:52-89: star of mass 1 at rest at the origin, n disc bodies of total mass 0.2 on
    near-Keplerian orbits between radii 1 and 10/2/1.2, |z| <= 5e-4-ish."""
    a = nb.disc(2000, seed=1)
    assert len(a) == 2001
    assert a["mass"][0] == 1.0 and not a["position"][0].any() and not a["velocity"][0].any()
    assert a["mass"][1:].sum() == pytest.approx(0.2, rel=1e-5)
    r = np.linalg.norm(a["position"][1:, :2], axis=1)
    assert r.min() >= 1.0 - 1e-6 and r.max() <= 10.0 / 2 / 1.2 + 1e-6
    assert np.abs(a["position"][1:, 2]).max() < 5e-3
    v = np.linalg.norm(a["velocity"][1:], axis=1)
    assert np.all(v * np.sqrt(r) > 0.99) and np.all(v * np.sqrt(r) < 1.1)   # v ~ sqrt(mu/a), 1 <= mu <= 1.2
    # clockwise seen from +z, as the reference sets it (vx = v sin(phi), vy = -v cos(phi))
    lz = a["position"][1:, 0] * a["velocity"][1:, 1] - a["position"][1:, 1] * a["velocity"][1:, 0]
    assert np.all(lz < 0)


def test_shard_range(nb):
    for n in (0, 1, 7, 8, 9, 65536, 1000003):
        for g in (1, 2, 3, 8):
            spans = [nb.shard_range(n, r, g) for r in range(g)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(g - 1))
            assert max(hi - lo for lo, hi in spans) <= -(-n // g)
