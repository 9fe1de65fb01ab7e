import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def nb():
    """The product: ctypes binding of libnbody_hip.so.  The test process itself never imports torch
    (torch bundles a second ROCm runtime; tests that need torch.distributed run it in child
    processes that import torch first)."""
    return graft.load_package()


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (checker only)."""
    o = graft.load_oracle()
    o.lib()
    return o


@pytest.fixture(scope="session")
def gpu(nb):
    if nb.device_count() < 1:
        pytest.fail("a test marked gpu ran without a HIP device: the HIP path has no fallback")
    return nb


@pytest.fixture(scope="session")
def gpu_tuning(gpu):
    """The tuning build of the same sources (libnbody_hip_tuning.so: experimental walks, in-kernel stamps), loaded beside the
    product as a second instance of the mirror.  Built by __graft_entry__.build(); a missing file is a failed build."""
    nbt = graft.load_package(tuning=True)
    assert nbt.is_tuning_build() and not gpu.is_tuning_build()
    return nbt


class Knob:
    """`knob.value = v`: every Simulation created from now on starts with that knob set (the mirror's tuning_defaults without a
    with-block; the library keeps its knobs per handle: include/nbody_hip.h nbody_set_tuning).  Setting the default removes it."""

    def __init__(self, nb, name, default):
        self.nb, self.name, self.default = nb, name, default

    @property
    def value(self):
        return self.nb._default_tuning.get(self.name, self.default)

    @value.setter
    def value(self, v):
        if v == self.default:
            self.nb._default_tuning.pop(self.name, None)
        else:
            self.nb._default_tuning[self.name] = int(v)


def particles(orc_or_dtype, pos, vel=None, mass=None):
    dt = orc_or_dtype
    pos = np.asarray(pos, dtype=np.float64)
    n = pos.shape[0]
    a = np.zeros(n, dtype=dt)
    a["position"] = pos
    if vel is not None:
        a["velocity"] = np.asarray(vel, dtype=np.float64)
    a["mass"] = 1.0 if mass is None else np.asarray(mass, dtype=np.float64)
    return a


def rel_err(got, ref):
    """max |got-ref| / max |ref| over a vector field."""
    scale = float(np.abs(ref).max())
    return float(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)).max()) / (scale if scale else 1.0)
