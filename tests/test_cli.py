"""The bench CLI that stands in for the reference's src/main.rs: same argv (`-t`, `-n`), same
workload (disc, dt=3e-2, g_soft=0.02, theta2=1.0, Barnes-Hut), same two output lines that
perf_benchmark.py's harness and the authors' notebook rely on."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "nbody-llm_amd", "nbody_cli")


def test_cli_is_built_and_fails_loudly_without_a_device(nb):
    assert os.path.exists(CLI), "run __graft_entry__.build()"
    if nb.device_count() > 0:
        pytest.skip("a HIP device is present")
    r = subprocess.run([CLI, "-t", "2", "-n", "100"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no HIP device" in r.stderr


def test_cli_rejects_unknown_flags():
    r = subprocess.run([CLI, "--bogus"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--tree", "device"], ["--tree", "host"], ["--method", "bf", "--ic", "plummer"], ["--dtype", "f64"], ["--dtype", "f64", "--tree", "device"],
                                   ["--dtype", "f64", "--method", "bf", "--ic", "plummer"]])
def test_cli_reference_argv_and_output_lines(gpu, extra):
    r = subprocess.run([CLI, "-t", "4", "-n", "3000", "--steps", "50"] + extra, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "Running simulation without rendering..." in out           # main.rs:111
    assert re.search(r"^Elapsed: [0-9.]+s$", out, re.M)                # main.rs:125
    m = re.search(r"^Performance: ([0-9.]+) steps/second$", out, re.M)  # main.rs:128
    assert m and float(m.group(1)) > 1.0
    left = int(re.search(r"Bodies left: (\d+)", out).group(1))
    assert 2500 <= left <= 3001


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--method", "bf", "--ic", "plummer", "--width", "3"], ["--method", "bh"], ["--dtype", "f64", "--method", "bh"]])
def test_cli_generic_integrator_on_the_host_equals_the_fused_device_one(gpu, tmp_path, extra):
    """The reference's trait is generic over its Integrator (src/shared.rs:99-104).  The device fuses the reference's
    LeapFrogIntegrator into its kernels; any other integrator runs on the host through the mirror's step_by_with (the
    unfused form of step_by: forces on the device, pre-/after-force and retain on the host).  With the leapfrog restated on
    the host and strict arithmetic the two runs end in the same bits -- bodies leaving the box on the way."""
    import numpy as np
    dumps = []
    for integ in ("device", "host"):
        f = str(tmp_path / f"{integ}.bin")
        r = subprocess.run([CLI, "-t", "2", "-n", "1200", "--steps", "12", "--math", "strict", "--integrator", integ, "--dump", f] + extra,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        dumps.append(np.fromfile(f, np.uint8))
    assert len(dumps[0]) > 0 and np.array_equal(dumps[0], dumps[1])
    if "--width" in extra:
        assert len(dumps[0]) < 1200 * 40      # some bodies left the tight box
