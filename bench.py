#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: pairwise interactions/s (+ steps/s) of the
N-body force-and-integrate step at N = 65 536, with the HBM/ALU roofline of the dominant kernel
and the CPU oracle timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload bf|bh] [--n BODIES]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one Simulation::step (half drift, retain, forces, kick + half drift) of ALL bodies.
Inputs are resident in HBM before the timed region.  With N ranks the bodies are split into N
contiguous index blocks; every rank exchanges its half-drifted positions once per step with an
RCCL all-gather issued by the library itself (torch.distributed/gloo is control plane only:
rendezvous, the ncclUniqueId broadcast, barriers and the max-over-ranks of the wall time).
BASELINE's metric is quoted at N = 65 536 on 1/2/4/8 GPUs, i.e. total work fixed: "strong".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_VALU_PEAK_TF = 157.3    # MI355X_MICROARCH.md: peak FP32 vector = f32 MFMA rate
FLOP_PER_INTERACTION = 20
BF_BYTES_PER_BODY = 32       # K2 alone: 16 B {x,y,z,m} read + 16 B acceleration written, per launch
BH_BYTES_PER_VISIT = 32      # K5: one 32-byte node record per opening test


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["bf", "bh"], default="bf",
                    help="bf = configs[1] (65 536-body brute force, the metric's config); bh = configs[2]")
    ap.add_argument("--n", type=int, default=65536, help="bodies over all GPUs")
    ap.add_argument("--math", choices=["fast", "strict"], default="fast")
    ap.add_argument("--theta", type=float, default=0.5, help="Barnes-Hut opening angle (theta2 = theta^2)")
    ap.add_argument("--tree", choices=["host", "device"], default="host",
                    help="Barnes-Hut octree build: host (north_star, bit-exact) or device (SURVEY F3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=20250523)
    return ap.parse_args()


def cpu_baseline(orc, ics, settings, box, workload):
    """The oracle (a port of the reference's CPU path) on this box's host cores, bounded to ~10-30 s."""
    center, width = box
    n = len(ics)
    if workload == "bf":
        # the reference loop is serial (brute_force.rs:70-81): one thread, one pass over all bodies
        # (~9 s at N = 65 536); larger N are cut to 65 536 bodies
        m = min(n, 65536)
        a = ics[:m].astype(orc.P32)
        t0 = time.perf_counter()
        orc.bf_update_forces(a, settings)
        dt = time.perf_counter() - t0
        threads = min(16, orc.hardware_threads())   # a 1-GPU box's share of the host cores
        b = ics[: min(n, 32768)].astype(orc.P32)
        t0 = time.perf_counter()
        orc.bf_update_forces_rows(b, settings, threads=threads)
        dt_mt = time.perf_counter() - t0
        mb = len(b)
        return {
            "value": m * (m - 1) / dt, "unit": "interactions/s", "cores": 1, "kind": "port",
            "sample": f"one update_forces pass over the first {m} bodies of the same Plummer set "
                      f"(serial symmetric pair loop as brute_force.rs:70-81, credited N(N-1) directed pairs), {dt:.1f} s",
            "threaded_context": {"value": mb * (mb - 1) / dt_mt, "cores": threads,
                                 "sample": f"row-wise form, {mb} bodies, {dt_mt:.1f} s"},
        }
    threads = min(16, orc.hardware_threads())       # a 1-GPU box's share of the host cores
    a = ics.astype(orc.P32)
    reps, acc = 3, 0
    t0 = time.perf_counter()
    for _ in range(reps):
        acc, _ = orc.bh_update_forces(a, settings, center, width, threads=threads)
    dt = (time.perf_counter() - t0) / reps
    return {
        "value": acc / dt, "unit": "interactions/s", "cores": threads, "kind": "port",
        "steps_per_sec": 1.0 / dt,
        "sample": f"{reps} update_forces passes (recursive build + threaded recursive walk as "
                  f"barnes_hut.rs:143-203,250-263) over all {len(a)} bodies, {dt:.2f} s each",
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched through torch.distributed.run (one rank per GPU)")
        args.gpus = world

    dist = None
    json_fd = None
    if "RANK" in os.environ and "MASTER_PORT" in os.environ:  # launched by torch.distributed.run (any N)
        # gloo and RCCL print connection/version banners on stdout: stdout is pointed at stderr for
        # the whole run and the one JSON line goes to the original descriptor at the end
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        # torch bundles its own ROCm runtime libraries under the same SONAMEs as /opt/rocm's; a
        # process must end up with ONE set, so torch goes first and libnbody_hip.so binds to what is
        # already loaded (the other order aborts at exit with a double free).  N = 1 never imports torch.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch.distributed as dist  # control plane only (gloo)
        dist.init_process_group("gloo", rank=rank, world_size=world)

    nb = graft.load_package()
    if nb.device_count() < 1:
        sys.exit("bench.py needs a HIP device: the engine has no CPU fallback")

    n = args.n
    box = ((0.0, 0.0, 0.0), 64.0)
    theta2 = args.theta * args.theta
    st = dict(g=1.0, g_soft=1e-2, dt=1e-3, theta2=theta2)
    ics = nb.plummer(n, seed=args.seed)
    method = nb.BRUTE_FORCE if args.workload == "bf" else nb.BARNES_HUT
    math_mode = nb.FAST if args.math == "fast" else nb.STRICT

    sim = nb.Simulation(ics, *box, method=method, math_mode=math_mode, capacity=n, device=local_rank,
                        rank=rank, world_size=world,
                        tree_build=nb.TREE_DEVICE if args.tree == "device" else nb.TREE_HOST)
    sim.settings = nb.Settings(**st)
    if dist is not None and (world > 1 or os.environ.get("NBODY_BENCH_FORCE_COMM")):
        ident = [nb.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        sim.comm_init(ident[0])
    sim.init()

    def barrier():
        sim.sync()
        if dist is not None:
            dist.barrier()
            sim.sync()

    sim.steps(args.warmup)
    sim.set_profiling(True)
    sim.reset_stats()
    barrier()
    t0 = time.perf_counter()
    sim.steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    stats = sim.stats()
    sim.set_profiling(False)
    n_after = sim.count_global() if world == 1 else None

    if dist is not None:
        import torch
        t = torch.tensor([elapsed, float(stats.interactions), stats.force_kernel_ms, float(stats.force_launches),
                          float(stats.node_visits), float(stats.force_kernel_interactions)], dtype=torch.float64)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        elapsed = max(float(g[0]) for g in gathered)
        interactions = sum(float(g[1]) for g in gathered)
        kernel_ms = max(float(g[2]) for g in gathered)   # the slowest rank's kernel time
        launches = float(gathered[0][3])
        visits = sum(float(g[4]) for g in gathered)
        k_inter = float(gathered[0][5])                  # rank 0's dominant-kernel interactions
    else:
        interactions, kernel_ms, launches, visits = float(stats.interactions), stats.force_kernel_ms, float(stats.force_launches), float(stats.node_visits)
        k_inter = float(stats.force_kernel_interactions)

    result = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = interactions / elapsed
        avg_kernel_ms = kernel_ms / max(1.0, launches)
        bodies_per_launch = n / world
        if args.workload == "bf":
            alg_bytes = BF_BYTES_PER_BODY * bodies_per_launch
            cross = os.environ.get("NBODY_CROSS_SYM", "1") != "0"
            kernel = ("k_bf_strict" if args.math == "strict" else
                      "k_bf_sym" if (world == 1 and n >= 8192) else
                      ("k_bf_cross" if cross else "k_bf_os") if (world > 1 and n // world >= 2048) else "k_bf_fast")
            # interactions the timed (dominant) launches evaluated; k_bf_sym leaves ~2 % (own and
            # opposite resident set) to the small companion kernel k_bf_sym_rest
            flops_per_launch = FLOP_PER_INTERACTION * k_inter / max(1.0, launches)
        else:
            alg_bytes = BH_BYTES_PER_VISIT * (visits / world) / max(1.0, launches) + 32 * bodies_per_launch
            kernel = "k_bh_walk"
            flops_per_launch = None
        achieved_gbs = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9 if avg_kernel_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.workload}.json")
        if os.path.exists(tpath) and world == 1 and n == 65536:
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {
            "bound": "hbm", "kernel": kernel, "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
            "avg_kernel_ms": avg_kernel_ms, "launches_timed": int(launches),
            "algorithmic_bytes_per_launch": alg_bytes,
        }
        if flops_per_launch is not None and avg_kernel_ms > 0:
            tf = flops_per_launch / (avg_kernel_ms * 1e-3) / 1e12
            # the all-pairs kernel is fp32-VALU bound, not HBM bound: this is
            # the fraction that says how good the kernel is
            roofline["alu"] = {"bound": "fp32-valu", "achieved": tf, "peak": FP32_VALU_PEAK_TF, "unit": "TFLOP/s",
                               "frac": tf / FP32_VALU_PEAK_TF, "flop_per_interaction": FLOP_PER_INTERACTION,
                               "interactions_per_launch": k_inter / max(1.0, launches)}
        if args.workload == "bh" and avg_kernel_ms > 0:
            # the walk's node records come out of L1/L2, not HBM (frac above can exceed 1): what bounds it is
            # the L1/TA pipeline serving divergent 16-byte gathers.  Cache-line
            # accesses per visit from the PMC pass kept in profiles/ (TCP_TOTAL_CACHE_ACCESSES, tools/pmc_bh.sh).
            per_visit = None
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_bh.json")))
                per_visit = pj["TCP_TOTAL_CACHE_ACCESSES_per_launch"] / 1.2e8
            except Exception:
                per_visit = None
            if per_visit:
                acc_per_cu_cycle = per_visit * (visits / world) / max(1.0, launches) / (avg_kernel_ms * 1e-3 * 2.4e9) / 256.0
                roofline["l1"] = {"bound": "l1-ta-pipeline", "achieved": acc_per_cu_cycle, "peak": 1.0,
                                  "unit": "cache-line accesses/cycle/CU", "frac": acc_per_cu_cycle,
                                  "accesses_per_visit": per_visit,
                                  "note": "node records are served by L1/L2 (the hbm fraction above counts algorithmic bytes); "
                                          "one cache-line access per cycle per CU nominal, cycles counted at 2.4 GHz"}
        result = {
            "metric": "pairwise_interactions_per_sec", "value": value, "unit": "interactions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "steps_per_sec": args.steps / elapsed,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": ("configs[1]: 65 536-body brute force" if (args.workload == "bf" and n == 65536) else
                             "configs[2]: 65 536-body Barnes-Hut theta=0.5" if (args.workload == "bh" and n == 65536) else
                             f"{args.workload} n={n}"),
                "n_bodies": n, "method": "brute_force" if args.workload == "bf" else "barnes_hut",
                "math": args.math, "ics": f"plummer seed={args.seed}", "dt": st["dt"], "g_soft": st["g_soft"],
                "theta2": theta2 if args.workload == "bh" else None, "box_width": box[1],
                "parallelism": "1 GPU" if world == 1 else
                               f"{world} index-block shards; per step one RCCL all-gather of positions and one "
                               f"send/recv round of partial sums (every pair between shards evaluated once)",
                "bodies_left_in_box": n_after,
            },
            "roofline": roofline,
        }
        if args.workload == "bh":
            result["bh"] = {"tree_build": args.tree, "tree_nodes": int(stats.tree_nodes), "node_visits_per_step": visits / args.steps,
                            "tree_build_ms_per_step": stats.tree_build_ms / args.steps,
                            "tree_copy_ms_per_step": stats.tree_copy_ms / args.steps}
    sim.close()

    if rank == 0:
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is timed at N = 1 only
            orc = graft.load_oracle()
            result["cpu_baseline"] = cpu_baseline(orc, ics, st, box, args.workload)
        else:
            result["cpu_baseline"] = None
        line = json.dumps(result)
        if json_fd is not None:
            os.write(json_fd, (line + "\n").encode())
        else:
            print(line, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
