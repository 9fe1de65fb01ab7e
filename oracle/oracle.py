"""ctypes wrapper of oracle/libnbody_oracle.so -- TEST INFRASTRUCTURE ONLY (parity unpinned: see
the header of oracle/nbody_oracle.cpp).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (nbody-llm_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnbody_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "nbody_oracle.cpp")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None
BUILD_FLAGS = "g++ -O3 -ffp-contract=off, no -march=native (built off-box; oracle/Makefile)"


def use_native() -> str:
    """bench.py's cpu_baseline leg: the oracle rebuilt for THIS machine (-march=native, same source, same
    -ffp-contract=off: same results) if a compiler is here and the library is not loaded yet.  Returns the flags in use."""
    global LIB_PATH, BUILD_FLAGS
    if _lib is None:
        native = os.path.join(_HERE, "libnbody_oracle_native.so")
        try:
            subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            C.CDLL(native)   # (loads on this CPU?)
            LIB_PATH = native
            BUILD_FLAGS = "g++ -O3 -march=native -ffp-contract=off, built on the box the number comes from (oracle/Makefile `native`)"
        except Exception:  # noqa: BLE001 -- no compiler on the box: the portable build
            pass
    return BUILD_FLAGS


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
    return _lib


def particle_dtype(ftype):
    f = np.dtype(ftype)
    return np.dtype([("position", f, 3), ("velocity", f, 3), ("acceleration", f, 3), ("mass", f)])


P32 = particle_dtype(np.float32)
P64 = particle_dtype(np.float64)


def _sfx(a: np.ndarray):
    if a.dtype == P32:
        return "f32", C.c_float, np.float32
    if a.dtype == P64:
        return "f64", C.c_double, np.float64
    raise TypeError(f"expected a PointParticle record array, got {a.dtype}")


def _arr(ct, vals):
    return (ct * len(vals))(*[float(v) for v in vals])


def _settings(ct, s):
    return _arr(ct, [s["g"], s["g_soft"], s["dt"], s["theta2"]])


def default_settings() -> dict:
    out = (C.c_float * 4)()
    lib().oracle_default_settings_f32(out)
    return dict(g=out[0], g_soft=out[1], dt=out[2], theta2=out[3])


def to_f64(a32: np.ndarray) -> np.ndarray:
    out = np.zeros(a32.shape[0], dtype=P64)
    for k in ("position", "velocity", "acceleration", "mass"):
        out[k] = a32[k]
    return out


def pre_force(a, dt):
    s, ct, _ = _sfx(a)
    getattr(lib(), f"oracle_pre_force_{s}")(C.c_void_p(a.ctypes.data), C.c_size_t(len(a)), ct(dt))


def after_force(a, dt):
    s, ct, _ = _sfx(a)
    getattr(lib(), f"oracle_after_force_{s}")(C.c_void_p(a.ctypes.data), C.c_size_t(len(a)), ct(dt))


def retain(a, center, width) -> np.ndarray:
    s, ct, _ = _sfx(a)
    fn = getattr(lib(), f"oracle_retain_{s}")
    fn.restype = C.c_size_t
    n = fn(C.c_void_p(a.ctypes.data), C.c_size_t(len(a)), _arr(ct, center), ct(width))
    return a[:n]


def bf_update_forces(a, settings):
    s, ct, _ = _sfx(a)
    getattr(lib(), f"oracle_bf_update_forces_{s}")(C.c_void_p(a.ctypes.data), C.c_size_t(len(a)), _settings(ct, settings))


def bf_update_forces_rows(a, settings, threads=1):
    s, ct, _ = _sfx(a)
    getattr(lib(), f"oracle_bf_update_forces_rows_{s}")(C.c_void_p(a.ctypes.data), C.c_size_t(len(a)),
                                                       _settings(ct, settings), C.c_int(threads))


def bf_update_forces_range(a, settings, row0, row1, threads=1):
    """Accelerations of bodies [row0,row1) only, against all bodies (same arithmetic as the rows form)."""
    s, ct, _ = _sfx(a)
    getattr(lib(), f"oracle_bf_update_forces_range_{s}")(C.c_void_p(a.ctypes.data), C.c_size_t(len(a)),
                                                        _settings(ct, settings), C.c_int(threads),
                                                        C.c_size_t(row0), C.c_size_t(row1))


def bf_step_by(a, settings, center, width, dt) -> np.ndarray:
    """One BruteForceSimulation::step_by; returns the (possibly shorter) body array view."""
    s, ct, _ = _sfx(a)
    fn = getattr(lib(), f"oracle_bf_step_by_{s}")
    fn.restype = C.c_size_t
    n = fn(C.c_void_p(a.ctypes.data), C.c_size_t(len(a)), _settings(ct, settings), _arr(ct, center), ct(width), ct(dt))
    return a[:n]


def bh_update_forces(a, settings, center, width, threads=1, leaf_mode=0):
    """Returns (accepted, visited) node counts.  leaf_mode 0: src/manual (a leaf failing the opening test
    contributes nothing); 1: the src/llm walk on the same tree (such a leaf is evaluated directly)."""
    s, ct, _ = _sfx(a)
    fn = getattr(lib(), f"oracle_bh_update_forces_mode_{s}")
    fn.restype = C.c_int
    acc, vis = C.c_uint64(0), C.c_uint64(0)
    rc = fn(C.c_void_p(a.ctypes.data), C.c_size_t(len(a)), _settings(ct, settings), _arr(ct, center), ct(width),
            C.c_int(threads), C.byref(acc), C.byref(vis), C.c_int(leaf_mode))
    if rc:
        raise RuntimeError(f"oracle_bh_update_forces rc={rc}")
    return acc.value, vis.value


def bh_step_by(a, settings, center, width, dt, threads=1, leaf_mode=0):
    """One BarnesHutSimulation::step_by; returns (array view, accepted, visited)."""
    s, ct, _ = _sfx(a)
    fn = getattr(lib(), f"oracle_bh_step_by_mode_{s}")
    fn.restype = C.c_size_t
    acc, vis, rc = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
    n = fn(C.c_void_p(a.ctypes.data), C.c_size_t(len(a)), _settings(ct, settings), _arr(ct, center), ct(width), ct(dt),
           C.c_int(threads), C.byref(acc), C.byref(vis), C.byref(rc), C.c_int(leaf_mode))
    if rc.value:
        raise RuntimeError(f"oracle_bh_step_by rc={rc.value}")
    return a[:n], acc.value, vis.value


def bh_build_tree(a, center, width) -> dict:
    s, ct, ft = _sfx(a)
    fn = getattr(lib(), f"oracle_bh_build_tree_{s}")
    fn.restype = C.c_long
    args = (C.c_void_p(a.ctypes.data), C.c_size_t(len(a)), _arr(ct, center), ct(width))
    m = fn(*args, None, None, None, None, None, C.c_size_t(0))
    if m < 0:
        raise RuntimeError(f"oracle_bh_build_tree rc={m}")
    com = np.zeros((m, 4), ft)
    w = np.zeros(m, ft)
    skip = np.zeros(m, np.int32)
    nchild = np.zeros(m, np.int32)
    leaf = np.zeros(m, np.int32)
    fn(*args, C.c_void_p(com.ctypes.data), C.c_void_p(w.ctypes.data), C.c_void_p(skip.ctypes.data),
       C.c_void_p(nchild.ctypes.data), C.c_void_p(leaf.ctypes.data), C.c_size_t(m))
    return dict(com_mass=com, width=w, skip=skip, nchild=nchild, leaf_body=leaf)


def bh_tree_cells(a, center, width) -> np.ndarray:
    """[n_nodes, 6] {min xyz, max xyz} of every node of the reference's tree, pre-order (what its renderer draws)."""
    s, ct, ft = _sfx(a)
    fn = getattr(lib(), f"oracle_bh_tree_cells_{s}")
    fn.restype = C.c_long
    args = (C.c_void_p(a.ctypes.data), C.c_size_t(len(a)), _arr(ct, center), ct(width))
    m = fn(*args, None, C.c_size_t(0))
    if m < 0:
        raise RuntimeError(f"oracle_bh_tree_cells rc={m}")
    out = np.zeros((m, 6), ft)
    fn(*args, C.c_void_p(out.ctypes.data), C.c_size_t(m))
    return out


def energy(a, g=1.0, g_soft=0.0, threads=0):
    """f64 (KE, PE) of a state."""
    s, _, _ = _sfx(a)
    if threads <= 0:
        threads = hardware_threads()
    ke, pe = C.c_double(0), C.c_double(0)
    getattr(lib(), f"oracle_energy_{s}")(C.c_void_p(a.ctypes.data), C.c_size_t(len(a)), C.c_double(g),
                                         C.c_double(g_soft), C.c_int(threads), C.byref(ke), C.byref(pe))
    return ke.value, pe.value


def hardware_threads() -> int:
    return int(lib().oracle_hardware_threads())
