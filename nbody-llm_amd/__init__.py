"""nbody_llm_amd -- ctypes binding of libnbody_hip.so (include/nbody_hip.h) and a thin host-side
mirror of the reference's `Simulation` trait (src/shared.rs:80-97) for tests and bench.py.

The directory is named `nbody-llm_amd`; load it under the module name `nbody_llm_amd` with
`__graft_entry__.load_package()` (a hyphen cannot be imported directly).

There is NO fallback: if libnbody_hip.so is missing this module raises at import, and without a
HIP device `Simulation(...)` raises NbodyError(NBODY_ERR_NO_DEVICE).  Nothing here touches
oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# NBODY_HIP_LIBRARY: another build of the same ABI -- libnbody_hip_tuning.so carries the experimental walks and the
# in-kernel stamps the release library leaves out (make -C nbody-llm_amd/csrc tuning; __graft_entry__.load_package(tuning=True))
LIB_PATH = os.environ.get("NBODY_HIP_LIBRARY") or os.path.join(_HERE, "libnbody_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C nbody-llm_amd/csrc` (or __graft_entry__.build()); "
        "there is no Python/CPU fallback for the HIP engine"
    )

lib = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)   # (two builds of the library may live in one process: nothing is resolved across them)

# ---- constants (include/nbody_hip.h) ----------------------------------------------------------
NBODY_OK = 0
NBODY_ERR_INVALID = -1
NBODY_ERR_HIP = -2
NBODY_ERR_CAPACITY = -3
NBODY_ERR_TREE_DEPTH = -4
NBODY_ERR_COMM = -5
NBODY_ERR_NO_DEVICE = -6
BRUTE_FORCE, BARNES_HUT = 0, 1
STRICT, FAST = 0, 1
LEAF_REFERENCE = 0   # src/manual: a leaf failing the opening test contributes nothing
LEAF_DIRECT = 1      # the src/llm walk on the same tree: such a leaf is evaluated directly
TREE_HOST, TREE_DEVICE, TREE_AUTO = 0, 1, 2   # AUTO: fast math -> device build, strict -> host build
COMM_ID_BYTES = 128

#: PointParticle<f32,3>, #[repr(C)] (src/shared.rs:151-158)
PARTICLE_DTYPE = np.dtype(
    [("position", "<f4", 3), ("velocity", "<f4", 3), ("acceleration", "<f4", 3), ("mass", "<f4")]
)
assert PARTICLE_DTYPE.itemsize == 40
#: PointParticle<f64,3> (the instantiation the reference's own driver uses, src/main.rs:52-105): NBODY_F64 handles
PARTICLE_DTYPE64 = np.dtype(
    [("position", "<f8", 3), ("velocity", "<f8", 3), ("acceleration", "<f8", 3), ("mass", "<f8")]
)
assert PARTICLE_DTYPE64.itemsize == 80
F32, F64 = 0, 1
SHARD_INDEX, SHARD_SPATIAL = 0, 1   # index blocks + all-gather | Morton-key ranges + halo exchange (Barnes-Hut, fast math)

#: every symbol include/nbody_hip.h declares (tests check the library exports all of them)
DECLARED_SYMBOLS = [
    "nbody_create", "nbody_destroy", "nbody_clone", "nbody_upload", "nbody_download", "nbody_count",
    "nbody_count_global", "nbody_add_point", "nbody_remove_point", "nbody_set_settings", "nbody_get_settings",
    "nbody_set_bounds", "nbody_init", "nbody_step_by", "nbody_steps", "nbody_update_forces", "nbody_elapsed",
    "nbody_sync", "nbody_set_profiling", "nbody_stats", "nbody_reset_stats", "nbody_energy", "nbody_tree_export",
    "nbody_last_error", "nbody_comm_unique_id", "nbody_comm_init", "nbody_local_range", "nbody_ic_plummer",
    "nbody_ic_disc", "nbody_host_build_tree", "nbody_abi_version", "nbody_device_count",
    "nbody_debug_step_begin", "nbody_debug_import_segment", "nbody_debug_step_forces",
    "nbody_debug_import_partials", "nbody_debug_step_end", "nbody_host_cross_plan",
    "nbody_set_settings_f64", "nbody_get_settings_f64", "nbody_set_bounds_f64", "nbody_step_by_f64", "nbody_elapsed_f64",
    "nbody_tree_export_f64", "nbody_ic_plummer_f64", "nbody_ic_disc_f64",
    "nbody_download_ids", "nbody_let_stats", "nbody_debug_let_phase", "nbody_debug_let_exchange", "nbody_debug_let_set_prune",
    "nbody_debug_let_bounds", "nbody_debug_let_set_balance",
    "nbody_comm_local_id", "nbody_comm_transport", "nbody_host_exchange_layout",
    "nbody_set_tuning", "nbody_get_tuning", "nbody_is_tuning_build", "nbody_tree_export_cells", "nbody_host_launch_plan",
]


class NbodyConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("method", C.c_int32), ("math_mode", C.c_int32), ("leaf_mode", C.c_int32),
        ("device", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32), ("host_threads", C.c_int32),
        ("capacity", C.c_uint64), ("tree_build", C.c_int32), ("dtype", C.c_int32), ("shard_mode", C.c_int32), ("reserved", C.c_int32),
    ]


class NbodyLetStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("steps", "bodies_migrated", "nodes_local", "nodes_global", "nodes_sent", "nodes_received",
                                          "bytes_sent", "bytes_allgather_equivalent")] + [("phase_ms", C.c_double * 5)] + \
               [(k, C.c_uint64) for k in ("host_syncs", "migrant_respills", "node_array_peak_bytes", "node_array_bytes")]


class NbodyStats(C.Structure):
    _fields_ = [
        ("steps", C.c_uint64), ("interactions", C.c_uint64), ("node_visits", C.c_uint64), ("tree_nodes", C.c_uint64),
        ("force_launches", C.c_uint64), ("force_kernel_interactions", C.c_uint64), ("force_kernel_ms", C.c_double), ("tree_build_ms", C.c_double),
        ("tree_copy_ms", C.c_double), ("exchange_ms", C.c_double),
    ]


_H = C.c_void_p
_f, _sz, _i = C.c_float, C.c_size_t, C.c_int
_pf = C.POINTER(C.c_float)


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_sig("nbody_create", _i, C.POINTER(NbodyConfig), C.POINTER(_H))
_sig("nbody_destroy", None, _H)
_sig("nbody_clone", _i, _H, C.POINTER(_H))
_sig("nbody_upload", _i, _H, C.c_void_p, _sz, _sz)
_sig("nbody_download", _i, _H, C.c_void_p, _sz, _sz, C.POINTER(_sz))
_sig("nbody_count", _i, _H, C.POINTER(_sz))
_sig("nbody_count_global", _i, _H, C.POINTER(_sz))
_sig("nbody_add_point", _i, _H, C.c_void_p)
_sig("nbody_remove_point", _i, _H, _sz)
_sig("nbody_set_settings", _i, _H, _f, _f, _f, _f)
_sig("nbody_get_settings", _i, _H, _pf, _pf, _pf, _pf)
_sig("nbody_set_bounds", _i, _H, _pf, _f)
_sig("nbody_init", _i, _H)
_sig("nbody_step_by", _i, _H, _f)
_sig("nbody_steps", _i, _H, _i)
_sig("nbody_update_forces", _i, _H)
_sig("nbody_elapsed", _i, _H, _pf)
_sig("nbody_sync", _i, _H)
_sig("nbody_set_profiling", _i, _H, _i)
_sig("nbody_stats", _i, _H, C.POINTER(NbodyStats))
_sig("nbody_reset_stats", _i, _H)
_sig("nbody_energy", _i, _H, C.POINTER(C.c_double), C.POINTER(C.c_double))
_sig("nbody_tree_export", _i, _H, C.c_void_p, C.c_void_p, C.c_void_p, _sz, C.POINTER(_sz))
_sig("nbody_tree_export_cells", _i, _H, C.c_void_p, C.c_void_p, _sz, C.POINTER(_sz))
_sig("nbody_last_error", C.c_char_p, _H)
_sig("nbody_comm_unique_id", _i, C.c_void_p)
_sig("nbody_comm_init", _i, _H, C.c_void_p)
_sig("nbody_comm_local_id", _i, C.c_void_p)
_sig("nbody_comm_transport", _i, _H, C.c_char_p, _sz)
_sig("nbody_local_range", _i, _H, C.POINTER(_sz), C.POINTER(_sz))
_sig("nbody_ic_plummer", _i, C.c_void_p, _sz, _sz, C.c_uint64)
_sig("nbody_ic_disc", _i, C.c_void_p, _sz, _sz, C.c_uint64)
_sig("nbody_host_build_tree", _i, C.c_void_p, _sz, _pf, _f, _i, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
     C.c_void_p, _sz, C.POINTER(_sz))
_sig("nbody_debug_step_begin", _i, _H, _f)
_sig("nbody_debug_import_segment", _i, _H, _H)
_sig("nbody_debug_step_forces", _i, _H, _f)
_sig("nbody_debug_import_partials", _i, _H, _H)
_sig("nbody_debug_step_end", _i, _H, _f)
_sig("nbody_host_cross_plan", _i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.c_void_p, C.POINTER(_i),
     C.c_void_p)
_d = C.c_double
_pd = C.POINTER(C.c_double)
_sig("nbody_set_settings_f64", _i, _H, _d, _d, _d, _d)
_sig("nbody_get_settings_f64", _i, _H, _pd, _pd, _pd, _pd)
_sig("nbody_set_bounds_f64", _i, _H, _pd, _d)
_sig("nbody_step_by_f64", _i, _H, _d)
_sig("nbody_elapsed_f64", _i, _H, _pd)
_sig("nbody_tree_export_f64", _i, _H, C.c_void_p, C.c_void_p, C.c_void_p, _sz, C.POINTER(_sz))
_sig("nbody_ic_plummer_f64", _i, C.c_void_p, _sz, _sz, C.c_uint64)
_sig("nbody_ic_disc_f64", _i, C.c_void_p, _sz, _sz, C.c_uint64)
_sig("nbody_download_ids", _i, _H, C.c_void_p, _sz, C.POINTER(_sz))
_sig("nbody_let_stats", _i, _H, C.POINTER(NbodyLetStats))
_sig("nbody_debug_let_phase", _i, _H, _i, _f)
_sig("nbody_debug_let_exchange", _i, _H, _H, _i)
_sig("nbody_debug_let_set_prune", _i, _H, _i)
_sig("nbody_debug_let_bounds", _i, _H, C.c_void_p)
_sig("nbody_debug_let_set_balance", _i, _H, _i)
_sig("nbody_host_exchange_layout", _i, C.c_void_p, _i, _i, C.c_longlong, _i, _sz, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_sz))
_sig("nbody_host_launch_plan", _i, _sz, _f, _i, C.POINTER(_i))
_sig("nbody_set_tuning", _i, _H, C.c_char_p, _i)
_sig("nbody_get_tuning", _i, _H, C.c_char_p, C.POINTER(_i))
_sig("nbody_is_tuning_build", _i)
_sig("nbody_abi_version", _i)
_sig("nbody_device_count", _i)


#: knobs every Simulation created from here on starts with (host-side convenience of this mirror; the library itself
#: keeps its knobs per handle): `with tuning_defaults(bh_walk_split=4): ...`
_default_tuning: dict = {}


class tuning_defaults:
    def __init__(self, **knobs):
        self.knobs = knobs

    def __enter__(self):
        self.saved = dict(_default_tuning)
        _default_tuning.update(self.knobs)
        return self

    def __exit__(self, *a):
        _default_tuning.clear()
        _default_tuning.update(self.saved)


def launch_plan(n_bodies: int, theta2: float = 0.25, fast_math: bool = True) -> dict:
    """The launch shapes the library derives from a body count with its default knobs (nbody_host_launch_plan)."""
    out = (C.c_int * 3)()
    rc = lib.nbody_host_launch_plan(int(n_bodies), float(theta2), int(bool(fast_math)), out)
    if rc != 0:
        raise NbodyError(rc, "nbody_host_launch_plan")
    return {"walk_bodies_per_lane": out[0], "walk_segments": out[1], "sym_bodies_per_lane": out[2]}


def is_tuning_build() -> bool:
    return bool(lib.nbody_is_tuning_build())


class NbodyError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"nbody error {code}: {msg}")
        self.code = code


def device_count() -> int:
    return int(lib.nbody_device_count())


def plummer(n: int, seed: int = 20250523, f64: bool = False) -> np.ndarray:
    """Plummer sphere in Henon units as PointParticle records (host side; generated in f64, rounded to f32 unless f64)."""
    if f64:
        out = np.zeros(n, dtype=PARTICLE_DTYPE64)
        rc = lib.nbody_ic_plummer_f64(out.ctypes.data, n, 80, seed)
        if rc:
            raise NbodyError(rc, "nbody_ic_plummer_f64")
        return out
    out = np.zeros(n, dtype=PARTICLE_DTYPE)
    rc = lib.nbody_ic_plummer(out.ctypes.data, n, 40, seed)
    if rc:
        raise NbodyError(rc, "nbody_ic_plummer")
    return out


def disc(n_disc: int, seed: int = 20250523, f64: bool = False) -> np.ndarray:
    """The reference's self-gravitating disc (src/main.rs:52-89): 1 star + n_disc bodies."""
    if f64:
        out = np.zeros(n_disc + 1, dtype=PARTICLE_DTYPE64)
        rc = lib.nbody_ic_disc_f64(out.ctypes.data, n_disc, 80, seed)
        if rc:
            raise NbodyError(rc, "nbody_ic_disc_f64")
        return out
    out = np.zeros(n_disc + 1, dtype=PARTICLE_DTYPE)
    rc = lib.nbody_ic_disc(out.ctypes.data, n_disc, 40, seed)
    if rc:
        raise NbodyError(rc, "nbody_ic_disc")
    return out


def host_build_tree(pos4: np.ndarray, center, width: float, threads: int = 1):
    """Host-only octree build (no device).  Returns dict(com_mass, width, skip, leaf_body, order)."""
    pos4 = np.ascontiguousarray(pos4, dtype=np.float32).reshape(-1, 4)
    n = pos4.shape[0]
    c = (C.c_float * 3)(*[float(x) for x in center])
    nn = C.c_size_t(0)
    rc = lib.nbody_host_build_tree(pos4.ctypes.data, n, c, width, threads, None, None, None, None, None, 0, C.byref(nn))
    if rc:
        raise NbodyError(rc, "nbody_host_build_tree")
    m = nn.value
    com = np.zeros((m, 4), np.float32)
    w = np.zeros(m, np.float32)
    skip = np.zeros(m, np.int32)
    body = np.zeros(m, np.int32)
    order = np.zeros(n, np.int32)
    rc = lib.nbody_host_build_tree(pos4.ctypes.data, n, c, width, threads, com.ctypes.data, w.ctypes.data,
                                   skip.ctypes.data, body.ctypes.data, order.ctypes.data, m, C.byref(nn))
    if rc:
        raise NbodyError(rc, "nbody_host_build_tree")
    return dict(com_mass=com, width=w, skip=skip, leaf_body=body, order=order)


@dataclass
class Settings:
    """SimulationSettings (src/shared.rs:61-78)."""
    g: float = 1.0
    g_soft: float = 0.0
    dt: float = 1e-3
    theta2: float = 0.5


class Simulation:
    """Host-side mirror of the reference's `Simulation` trait over the C ABI.

    `Simulation(points, center, width, method=...)` is `Simulation::new(points, LeapFrogIntegrator,
    Bounds::new(center, width))`; methods keep the trait's names and meaning.
    """

    def __init__(self, points: np.ndarray, center=(0.0, 0.0, 0.0), width: float = 1.0, *, method: int = BRUTE_FORCE,
                 math_mode: int = STRICT, capacity: int | None = None, device: int = -1, rank: int = 0,
                 world_size: int = 1, host_threads: int = 0, tree_build: int = TREE_AUTO, leaf_mode: int = LEAF_REFERENCE,
                 f64: bool | None = None, shard_mode: int = SHARD_INDEX, tuning: dict | None = None, _handle=None, _f64: bool = False):
        self._h = _H()
        self.f64 = bool(_f64)
        self.rank, self.world_size = int(rank), int(world_size)
        if _handle is not None:
            self._h = _handle
            return
        # F = f64 when the records are PointParticle<f64,3> (or asked for): NBODY_F64 handle, 80-byte records
        self.f64 = bool(f64) if f64 is not None else (getattr(points, "dtype", None) == PARTICLE_DTYPE64)
        self.dtype = PARTICLE_DTYPE64 if self.f64 else PARTICLE_DTYPE
        points = np.ascontiguousarray(points, dtype=self.dtype)
        cfg = NbodyConfig(C.sizeof(NbodyConfig), method, math_mode, leaf_mode, device, rank, world_size,
                          host_threads, int(capacity if capacity is not None else max(1, points.shape[0])), tree_build,
                          F64 if self.f64 else F32, shard_mode, 0)
        rc = lib.nbody_create(C.byref(cfg), C.byref(self._h))
        if rc:
            self._h = _H()
            raise NbodyError(rc, (lib.nbody_last_error(None) or b"").decode())
        for name, value in {**_default_tuning, **(tuning or {})}.items():
            self.set_tuning(name, value)
        self.set_bounds(center, width)
        self._check(lib.nbody_upload(self._h, points.ctypes.data, points.shape[0], self.dtype.itemsize))

    # -- plumbing
    def _check(self, rc: int):
        if rc:
            raise NbodyError(rc, (lib.nbody_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            lib.nbody_destroy(self._h)
            self._h = _H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- the trait surface (shared.rs:80-97)
    def init(self):
        self._check(lib.nbody_init(self._h))

    def step(self):
        self._check(lib.nbody_steps(self._h, 1))

    def steps(self, k: int):
        self._check(lib.nbody_steps(self._h, int(k)))

    def step_by(self, dt: float):
        self._check(lib.nbody_step_by_f64(self._h, float(dt)) if self.f64 else lib.nbody_step_by(self._h, float(dt)))

    def update_forces(self):
        self._check(lib.nbody_update_forces(self._h))

    def add_point(self, particle: np.ndarray):
        p = np.ascontiguousarray(particle, dtype=self.dtype).reshape(1)
        self._check(lib.nbody_add_point(self._h, p.ctypes.data))

    def remove_point(self, index: int):
        self._check(lib.nbody_remove_point(self._h, int(index)))

    def get_points(self) -> np.ndarray:
        n = C.c_size_t(0)
        self._check(lib.nbody_count(self._h, C.byref(n)))
        out = np.zeros(n.value, dtype=self.dtype)
        self._check(lib.nbody_download(self._h, out.ctypes.data, n.value, self.dtype.itemsize, C.byref(n)))
        return out[: n.value]

    def __len__(self) -> int:
        n = C.c_size_t(0)
        self._check(lib.nbody_count(self._h, C.byref(n)))
        return int(n.value)

    def count_global(self) -> int:
        n = C.c_size_t(0)
        self._check(lib.nbody_count_global(self._h, C.byref(n)))
        return int(n.value)

    def elapsed(self) -> float:
        if self.f64:
            d = C.c_double(0)
            self._check(lib.nbody_elapsed_f64(self._h, C.byref(d)))
            return float(d.value)
        v = C.c_float(0)
        self._check(lib.nbody_elapsed(self._h, C.byref(v)))
        return float(v.value)

    @property
    def settings(self) -> Settings:
        if self.f64:
            g, e, dt, t2 = C.c_double(), C.c_double(), C.c_double(), C.c_double()
            self._check(lib.nbody_get_settings_f64(self._h, C.byref(g), C.byref(e), C.byref(dt), C.byref(t2)))
            return Settings(g.value, e.value, dt.value, t2.value)
        g, e, dt, t2 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        self._check(lib.nbody_get_settings(self._h, C.byref(g), C.byref(e), C.byref(dt), C.byref(t2)))
        return Settings(g.value, e.value, dt.value, t2.value)

    @settings.setter
    def settings(self, s: Settings):
        if self.f64:
            self._check(lib.nbody_set_settings_f64(self._h, float(s.g), float(s.g_soft), float(s.dt), float(s.theta2)))
        else:
            self._check(lib.nbody_set_settings(self._h, float(s.g), float(s.g_soft), float(s.dt), float(s.theta2)))

    def set_bounds(self, center, width: float):
        if self.f64:
            self._check(lib.nbody_set_bounds_f64(self._h, (C.c_double * 3)(*[float(x) for x in center]), float(width)))
        else:
            self._check(lib.nbody_set_bounds(self._h, (C.c_float * 3)(*[float(x) for x in center]), float(width)))

    def clone(self) -> "Simulation":
        h = _H()
        rc = lib.nbody_clone(self._h, C.byref(h))
        if rc:
            raise NbodyError(rc, (lib.nbody_last_error(None) or b"").decode())
        twin = Simulation(None, rank=self.rank, world_size=self.world_size, _handle=h, _f64=self.f64)
        twin.dtype = self.dtype
        return twin

    # -- diagnostics / multi-GPU
    def sync(self):
        self._check(lib.nbody_sync(self._h))

    def set_profiling(self, on):
        """False / True: HIP events around every force-kernel launch; an int k > 1: around every k-th."""
        self._check(lib.nbody_set_profiling(self._h, int(on)))

    def stats(self) -> NbodyStats:
        s = NbodyStats()
        self._check(lib.nbody_stats(self._h, C.byref(s)))
        return s

    def reset_stats(self):
        self._check(lib.nbody_reset_stats(self._h))

    def energy(self) -> tuple[float, float]:
        ke, pe = C.c_double(), C.c_double()
        self._check(lib.nbody_energy(self._h, C.byref(ke), C.byref(pe)))
        return float(ke.value), float(pe.value)

    def tree(self):
        n = C.c_size_t(0)
        export = lib.nbody_tree_export_f64 if self.f64 else lib.nbody_tree_export
        ft = np.float64 if self.f64 else np.float32
        self._check(export(self._h, None, None, None, 0, C.byref(n)))
        m = n.value
        com = np.zeros((m, 4), ft)
        w = np.zeros(m, ft)
        skip = np.zeros(m, np.int32)
        self._check(export(self._h, com.ctypes.data, w.ctypes.data, skip.ctypes.data, m, C.byref(n)))
        return dict(com_mass=com, width=w, skip=skip)

    def tree_cells(self):
        """(min_max [n, 6] f32, depth [n] i32) of every node of the last tree, pre-order: what the reference's renderer draws."""
        n = C.c_size_t(0)
        self._check(lib.nbody_tree_export_cells(self._h, None, None, 0, C.byref(n)))
        mm = np.zeros((n.value, 6), np.float32)
        depth = np.zeros(n.value, np.int32)
        self._check(lib.nbody_tree_export_cells(self._h, mm.ctypes.data, depth.ctypes.data, n.value, C.byref(n)))
        return mm, depth

    def download_ids(self) -> np.ndarray:
        """NBODY_SHARD_SPATIAL: index in the uploaded vector of every body get_points() returns, in the same order."""
        n = C.c_size_t(0)
        self._check(lib.nbody_count(self._h, C.byref(n)))
        ids = np.zeros(n.value, np.int32)
        self._check(lib.nbody_download_ids(self._h, ids.ctypes.data, n.value, C.byref(n)))
        return ids[: n.value]

    def let_stats(self) -> NbodyLetStats:
        s = NbodyLetStats()
        self._check(lib.nbody_let_stats(self._h, C.byref(s)))
        return s

    def let_bounds(self) -> np.ndarray:
        """NBODY_SHARD_SPATIAL: the key-range bounds the next classification uses ([world_size + 1] uint64)."""
        out = np.zeros(self.world_size + 1, np.uint64)
        self._check(lib.nbody_debug_let_bounds(self._h, out.ctypes.data))
        return out

    def set_balance(self, by_work: bool):
        """NBODY_SHARD_SPATIAL: redraw the bounds at the quantiles of the last walk's visit counts (default) or of the body count."""
        self._check(lib.nbody_debug_let_set_balance(self._h, int(bool(by_work))))

    def set_prune(self, on: bool):
        self._check(lib.nbody_debug_let_set_prune(self._h, int(bool(on))))

    def local_range(self) -> tuple[int, int]:
        a, b = C.c_size_t(0), C.c_size_t(0)
        self._check(lib.nbody_local_range(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def comm_init(self, id_bytes: bytes):
        buf = C.create_string_buffer(bytes(id_bytes), COMM_ID_BYTES)
        self._check(lib.nbody_comm_init(self._h, buf))

    def set_tuning(self, name: str, value: int):
        """One launch-shape / scheme knob of this handle (include/nbody_hip.h nbody_set_tuning)."""
        self._check(lib.nbody_set_tuning(self._h, name.encode(), int(value)))

    def get_tuning(self, name: str) -> int:
        v = C.c_int(0)
        self._check(lib.nbody_get_tuning(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def comm_transport(self) -> str:
        buf = C.create_string_buffer(16)
        self._check(lib.nbody_comm_transport(self._h, buf, 16))
        return buf.value.decode()


def sharded_step(sims: list, dt: float | None = None):
    """One step of a world of len(sims) shards living in this process (test harness: what the RCCL
    all-gather of positions and the send/recv round of partial sums perform is done with
    device-to-device copies)."""
    dts = [float(s.settings.dt if dt is None else dt) for s in sims]
    for s, d in zip(sims, dts):
        s._check(lib.nbody_debug_step_begin(s._h, d))
    for s in sims:
        for peer in sims:
            if peer is not s:
                s._check(lib.nbody_debug_import_segment(s._h, peer._h))
    for s, d in zip(sims, dts):
        s._check(lib.nbody_debug_step_forces(s._h, d))
    for s in sims:
        for peer in sims:
            if peer is not s:
                s._check(lib.nbody_debug_import_partials(s._h, peer._h))
    for s, d in zip(sims, dts):
        s._check(lib.nbody_debug_step_end(s._h, d))


def spatial_step(sims: list, dt: float | None = None, forces_only: bool = False):
    """One step (or, forces_only, one update_forces) of a world of len(sims) NBODY_SHARD_SPATIAL handles living in this
    process: the five phases of nbody_let.cpp with the four RCCL exchanges done as device-to-device copies."""
    dts = [float(s.settings.dt if dt is None else dt) for s in sims]
    first, last = (10, 14) if forces_only else (0, 4)
    for phase in (first, 1, 2, 3, last):
        for s, d in zip(sims, dts):
            s._check(lib.nbody_debug_let_phase(s._h, phase, d))
        if phase == last:
            break
        which = {first: 0, 1: 1, 2: 2, 3: 3}[phase]
        for s in sims:
            for peer in sims:
                if peer is not s:
                    s._check(lib.nbody_debug_let_exchange(s._h, peer._h, which))


def spatial_gather(sims: list, n: int):
    """The world's bodies back in the uploaded vector's order: (records, indices that survived)."""
    parts, ids = [], []
    for s in sims:
        parts.append(s.get_points())
        ids.append(s.download_ids())
    rec = np.concatenate(parts)
    idx = np.concatenate(ids)
    order = np.argsort(idx, kind="stable")
    return rec[order], idx[order]


def host_cross_plan(rank: int, world: int, seg_cap: int, n_own: int) -> dict:
    """Host-only: the plan of the symmetric scheme across shards for one rank."""
    ipt, a, npart, nrecv = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    parts = np.zeros((8, 5), np.int32)
    recv = np.zeros(8, np.int32)
    rc = lib.nbody_host_cross_plan(rank, world, seg_cap, n_own, C.byref(ipt), C.byref(a), C.byref(npart),
                                   parts.ctypes.data, C.byref(nrecv), recv.ctypes.data)
    if rc:
        raise NbodyError(rc, "nbody_host_cross_plan")
    return dict(ipt=ipt.value, n_sets=a.value, parts=parts[: npart.value].copy(), recv_from=recv[: nrecv.value].copy())


def host_exchange_layout(matrix: np.ndarray, rank: int, clamp: int, packed_send: bool, send_stride: int = 0) -> dict:
    """Host-only: message offsets and counts of a variable-size round of the spatial step, for one rank."""
    m = np.ascontiguousarray(matrix, dtype=np.int32)
    G = m.shape[0]
    arr = {k: np.zeros(G, np.uint64) for k in ("out_at", "n_out", "in_at", "n_in")}
    tot = C.c_size_t(0)
    rc = lib.nbody_host_exchange_layout(m.ctypes.data, G, rank, int(clamp), int(bool(packed_send)), int(send_stride), arr["out_at"].ctypes.data,
                                        arr["n_out"].ctypes.data, arr["in_at"].ctypes.data, arr["n_in"].ctypes.data, C.byref(tot))
    if rc:
        raise NbodyError(rc, "nbody_host_exchange_layout")
    return dict({k: v.astype(np.int64) for k, v in arr.items()}, total_in=int(tot.value))


def comm_unique_id() -> bytes:
    """Id of an RCCL communicator (NBODY_TRANSPORT=ipc in the environment: of the one-device transport)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib.nbody_comm_unique_id(buf)
    if rc:
        raise NbodyError(rc, (lib.nbody_last_error(None) or b"").decode())
    return buf.raw


def comm_local_id() -> bytes:
    """Id of the one-device transport: ranks (processes or threads) that share a GPU."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib.nbody_comm_local_id(buf)
    if rc:
        raise NbodyError(rc, (lib.nbody_last_error(None) or b"").decode())
    return buf.raw


def shard_range(n: int, rank: int, world_size: int) -> tuple[int, int]:
    """Index block [lo, hi) that nbody_upload gives `rank` (contiguous blocks of ceil(n/G))."""
    blk = (n + world_size - 1) // world_size
    lo = min(n, rank * blk)
    return lo, min(n, lo + blk)
