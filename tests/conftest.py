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
