"""ranks.py -- run a world of G ranks through the PRODUCTION step (nbody_step_by / nbody_steps with the library's own
exchanges), each rank a fresh process (or a thread of one), with a stdlib control plane (rendezvous.py).

    parent:  results = run_world(cfg, ranks_per_process=1, timeout=180)
    child:   python ranks.py <cfg.json> <r0[,r1,...]>

With cfg["transport"] == "ipc" the ranks share ONE device (csrc/transport_ipc.hip): that is how the multi-rank control
flow -- count matrices, variable-size rounds, stream ordering, host synchronisations -- runs on a one-GPU box, where
RCCL refuses to put two ranks on one device.  With "rccl", rank r uses device r (one process per GPU).

cfg (JSON): world, address (socket path), out (directory), transport, device (ipc: the shared device),
  sim {method bf|bh, math fast|strict, shard index|spatial, tree auto|host|device, leaf reference|direct, tuning {knob: value}},
  ics {kind plummer|disc, n, seed, mass_jitter (seed or null)}, box [[cx, cy, cz], width], settings {g, g_soft, dt, theta2},
  schedule [["steps", k] | ["step_by", dt] | ["update_forces"] | ["settings", {...}] | ["sync"]], env {NAME: value},
  env_by_rank {"r": {NAME: value}}.
Every rank leaves out/rank<r>.npz (its bodies; spatial shards: + their indices in the uploaded vector) and
out/rank<r>.json (counts, statistics, wall time of the schedule); a failing rank leaves out/rank<r>.err.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile
import threading
import time
import traceback

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def _package():
    if "nbody_llm_amd" in sys.modules:
        return sys.modules["nbody_llm_amd"]
    root = os.path.dirname(_HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    import __graft_entry__ as graft
    return graft.load_package()


def make_ics(nb, ics: dict) -> np.ndarray:
    n, seed, f64 = int(ics["n"]), int(ics.get("seed", 20250523)), bool(ics.get("f64", False))
    pts = nb.disc(n, seed=seed, f64=f64) if ics.get("kind", "plummer") == "disc" else nb.plummer(n, seed=seed, f64=f64)
    if ics.get("mass_jitter") is not None:
        pts["mass"] *= np.random.default_rng(int(ics["mass_jitter"])).uniform(0.5, 1.5, len(pts)).astype(pts["mass"].dtype)
    if ics.get("velocity_scale") is not None:
        pts["velocity"] *= pts["velocity"].dtype.type(ics["velocity_scale"])
    return pts


def make_sim(nb, cfg: dict, points: np.ndarray, rank: int, world: int, device: int):
    sim_cfg = cfg["sim"]
    center, width = cfg["box"]
    return nb.Simulation(
        points, tuple(center), float(width),
        method=nb.BARNES_HUT if sim_cfg.get("method", "bf") == "bh" else nb.BRUTE_FORCE,
        math_mode=nb.FAST if sim_cfg.get("math", "fast") == "fast" else nb.STRICT,
        capacity=int(sim_cfg.get("capacity", len(points))), device=device, rank=rank, world_size=world,
        tree_build={"auto": nb.TREE_AUTO, "host": nb.TREE_HOST, "device": nb.TREE_DEVICE}[sim_cfg.get("tree", "auto")],
        leaf_mode=nb.LEAF_DIRECT if sim_cfg.get("leaf", "reference") == "direct" else nb.LEAF_REFERENCE,
        shard_mode=nb.SHARD_SPATIAL if sim_cfg.get("shard", "index") == "spatial" else nb.SHARD_INDEX,
        tuning=sim_cfg.get("tuning"))


def run_schedule(nb, sim, schedule, reattach=None):
    """Runs the schedule; returns the simulation it ended with (a "clone" entry replaces it by its clone, whose communicator
    `reattach(clone)` sets up again -- nbody_clone does not carry one over)."""
    for item in schedule:
        op = item[0]
        if op == "add_point":        # [x, y, z, vx, vy, vz, mass]: Vec::push (collective in a sharded world)
            p = np.zeros(1, nb.PARTICLE_DTYPE)
            p["position"], p["velocity"], p["mass"] = item[1][:3], item[1][3:6], item[1][6]
            sim.add_point(p)
        elif op == "remove_point":   # Vec::swap_remove
            sim.remove_point(int(item[1]))
        elif op == "clone":
            twin = sim.clone()
            if reattach is not None:
                reattach(twin)
            sim.close()
            sim = twin
        elif op == "steps":
            sim.steps(int(item[1]))
        elif op == "step_by":
            sim.step_by(float(item[1]))
        elif op == "update_forces":
            sim.update_forces()
        elif op == "settings":
            sim.settings = nb.Settings(**item[1])
        elif op == "sync":
            sim.sync()
        else:
            raise ValueError(f"unknown schedule entry {item}")
    return sim


def _rank_main(cfg: dict, rank: int, failed: list) -> None:
    out = cfg["out"]
    try:
        nb = _package()
        from nbody_llm_amd.rendezvous import Rendezvous
        world = int(cfg["world"])
        ipc = cfg.get("transport", "ipc") == "ipc"
        device = int(cfg.get("device", 0)) if ipc else rank
        rdzv = Rendezvous(rank, world, cfg["address"], timeout=float(cfg.get("timeout", 120.0)))
        points = make_ics(nb, cfg["ics"])
        sim = make_sim(nb, cfg, points, rank, world, device)
        sim.settings = nb.Settings(**cfg["settings"])
        ident = rdzv.bcast_bytes((nb.comm_local_id() if ipc else nb.comm_unique_id()) if rank == 0 else None)
        sim.comm_init(ident)
        sim.init()
        rdzv.barrier()

        def reattach(twin):
            twin.rank, twin.world_size = rank, world
            twin.comm_init(rdzv.bcast_bytes((nb.comm_local_id() if ipc else nb.comm_unique_id()) if rank == 0 else None))

        t0 = time.perf_counter()
        sim = run_schedule(nb, sim, cfg["schedule"], reattach)
        sim.sync()
        wall = time.perf_counter() - t0
        pts = sim.get_points()
        arrays = {"points": pts.view(np.uint8)}
        if cfg["sim"].get("shard", "index") == "spatial":
            arrays["ids"] = sim.download_ids()
        st = sim.stats()
        meta = {"rank": rank, "f64": bool(sim.f64), "count": int(len(pts)), "count_global": int(sim.count_global()), "wall_s": wall, "elapsed": sim.elapsed(),
                "transport": sim.comm_transport(), "steps": int(st.steps), "interactions": int(st.interactions),
                "node_visits": int(st.node_visits), "tree_nodes": int(st.tree_nodes), "local_range": list(sim.local_range())}
        if cfg["sim"].get("shard", "index") == "spatial":
            ls = sim.let_stats()
            meta["let"] = {k: int(getattr(ls, k)) for k in ("steps", "bodies_migrated", "nodes_local", "nodes_global", "nodes_sent", "nodes_received", "bytes_sent", "host_syncs",
                                                                "migrant_respills", "node_array_peak_bytes", "node_array_bytes")}
        rdzv.barrier()   # nobody leaves (and takes its window away) while a peer may still be inside an exchange
        sim.close()
        np.savez(os.path.join(out, f"rank{rank}.npz"), **arrays)
        with open(os.path.join(out, f"rank{rank}.json"), "w") as f:
            json.dump(meta, f)
        rdzv.close()
    except BaseException:  # noqa: BLE001 -- whatever it is goes into the rank's .err file; the parent reports it
        failed.append(rank)
        with open(os.path.join(out, f"rank{rank}.err"), "w") as f:
            f.write(traceback.format_exc())


def _child(argv) -> int:
    with open(argv[1]) as f:
        cfg = json.load(f)
    os.environ.update({k: str(v) for k, v in cfg.get("env", {}).items()})
    ranks = [int(x) for x in argv[2].split(",")]
    for r in ranks:   # (per-rank overrides: how the tests make ranks disagree)
        os.environ.update({k: str(v) for k, v in cfg.get("env_by_rank", {}).get(str(r), {}).items()})
    failed: list = []
    if len(ranks) > 1:
        # rank threads of one process: every stream needs a hardware queue of its own -- two streams folded onto one
        # queue run in submission order, and a device-side wait of one rank would then sit in front of the very kernel
        # of the other rank it is waiting for (the HIP runtime's default is four queues per process)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(2 * len(ranks) + 2))
    _package()   # (before any rank thread starts: a half-imported module is visible to other threads)
    if len(ranks) == 1:
        _rank_main(cfg, ranks[0], failed)
    else:   # several ranks of the world as threads of this process (the transport treats them like processes)
        threads = [threading.Thread(target=_rank_main, args=(cfg, r, failed)) for r in ranks]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    return 1 if failed else 0


def run_world(cfg: dict, ranks_per_process: int = 1, timeout: float = 180.0) -> list:
    """Start the world described by cfg as child processes, wait (bounded), return per rank {points, ids?, **meta}.
    A rank that fails or hangs is killed by PID and reported with what it wrote."""
    out = cfg["out"]
    os.makedirs(out, exist_ok=True)
    cfg = dict(cfg)
    # (a unix socket path holds ~100 characters: not under a deep test directory)
    cfg.setdefault("address", os.path.join(tempfile.gettempdir(), f"nbody_world_{os.getpid()}_{time.monotonic_ns() & 0xffffffff:x}.sock"))
    cfg.setdefault("timeout", min(timeout, 120.0))
    path = os.path.join(out, "world.json")
    with open(path, "w") as f:
        json.dump(cfg, f)
    world = int(cfg["world"])
    groups = [list(range(a, min(world, a + ranks_per_process))) for a in range(0, world, ranks_per_process)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for g in groups:
        log = open(os.path.join(out, f"proc{g[0]}.log"), "w")
        procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__), path, ",".join(map(str, g))], stdout=log, stderr=subprocess.STDOUT, env=env), log, g))
    deadline = time.monotonic() + timeout
    stuck = []
    for p, log, g in procs:
        try:
            p.wait(timeout=max(0.1, deadline - time.monotonic()))
        except subprocess.TimeoutExpired:
            stuck.append(g)
    for p, log, g in procs:
        if p.poll() is None:
            p.kill()
            p.wait()
        log.close()
    problems = []
    for p, log, g in procs:
        for r in g:
            err = os.path.join(out, f"rank{r}.err")
            if os.path.exists(err):
                problems.append(f"rank {r}:\n{open(err).read()}")
        if p.returncode != 0 and not any(os.path.exists(os.path.join(out, f"rank{r}.err")) for r in g):
            tail = open(os.path.join(out, f"proc{g[0]}.log")).read()[-2000:]
            problems.append(f"process of ranks {g} ended with {p.returncode}{' (killed: no result in time)' if g in stuck else ''}:\n{tail}")
    if problems:
        raise RuntimeError("multi-rank run failed:\n" + "\n".join(problems))
    nb = _package()
    results = []
    for r in range(world):
        z = np.load(os.path.join(out, f"rank{r}.npz"))
        with open(os.path.join(out, f"rank{r}.json")) as f:
            meta = json.load(f)
        meta["points"] = z["points"].view(nb.PARTICLE_DTYPE64 if meta.get("f64") else nb.PARTICLE_DTYPE)
        if "ids" in z:
            meta["ids"] = z["ids"]
        results.append(meta)
    return results


def gather_world(results: list) -> np.ndarray:
    """The world's bodies in the uploaded vector's order (index blocks: concatenation; spatial shards: by id)."""
    pts = np.concatenate([r["points"] for r in results])
    if "ids" in results[0]:
        ids = np.concatenate([r["ids"] for r in results])
        return pts[np.argsort(ids, kind="stable")]
    return pts


if __name__ == "__main__":
    sys.exit(_child(sys.argv))
