#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: pairwise interactions/s (+ steps/s) of the
N-body force-and-integrate step at N = 65 536, with the roofline of the dominant kernel and the CPU
oracle timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload bf|bh] [--n BODIES]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one Simulation::step (half drift, retain, forces, kick + half drift) of ALL bodies.
Inputs are resident in HBM before the timed region.  The default line is configs[1] (65 536-body
brute force, the configuration the metric is quoted on); on one GPU it also carries a `bh` object with
configs[2] (65 536-body Barnes-Hut, theta = 0.5) run both ways -- octree built on the host every step
(north_star's configuration) and built on the device -- each with its own step time, build / copy /
walk split, node visits and roofline.  With N ranks the bodies are split into N contiguous index
blocks; every rank exchanges its half-drifted positions once per step with an RCCL all-gather issued by
the library itself.  The control plane -- the 128-byte communicator id from rank 0, barriers, the
max-over-ranks of the wall time -- is nbody-llm_amd/rendezvous.py (stdlib sockets): a rank process never
imports torch, so libnbody_hip.so runs on the /opt/rocm HIP and RCCL it was compiled against (torch
bundles its own, older, under the same SONAMEs); `torch.distributed.run` is only the launcher that sets
RANK / WORLD_SIZE / MASTER_PORT.  BASELINE's metric is quoted at N = 65 536 on 1/2/4/8 GPUs, i.e. total
work fixed: "strong".  A one-GPU box rehearses the N > 1 path with NBODY_BENCH_DEVICE=0 NBODY_TRANSPORT=ipc
(all ranks on device 0 over the library's one-device transport; RCCL refuses two ranks on one device).

What bounds the kernels: the all-pairs kernel is fp32-VALU bound (O(N) data for
O(N^2) arithmetic), so `roofline.bound` is "fp32-valu" and the HBM fraction BASELINE asks for rides
along as `roofline.hbm`; the tree walk is bound by the L1/TA pipeline serving divergent 16-byte
gathers (its node records come out of L1/L2: HBM traffic is ~2 % of peak), so its `roofline.bound` is
"l1-ta" against the ceiling measured by tools/microbench_gather.hip (profiles/r02_gather_ceiling.json).
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
FP32_VALU_PEAK_TF = 157.3    # MI355X_MICROARCH.md: peak FP32 vector = 256 CUs x 256 flop/cycle x 2.4 GHz
FLOP_PER_INTERACTION = 20    # the convention rates are quoted in
FLOP_EXECUTED_SYM = 13       # what k_bf_sym issues: 17 VALU ops = 26 flop per UNORDERED pair = 2 interactions
BF_BYTES_PER_BODY = 32       # K2 alone: 16 B {x,y,z,m} read + 16 B acceleration written, per launch
BH_BYTES_PER_BODY = 32       # K5: 16 B position read + 16 B acceleration written (node records: see l1-ta)
CLOCK_HZ = 2.4e9             # nominal shader clock the per-cycle figures are quoted at
N_CU = 256

# which parity tests cover the kernel a given line times (say so in the line)
PARITY = {
    ("bf", "fast"): "fast math, tolerance not bit-exactness: tests/test_bf_gpu.py (acc <= 1e-5 of max|acc| vs the f32 oracle at "
                    "N = 65 536; <= 3e-5 vs the bit-exact strict kernel; pos <= 1e-4 after 100 steps; |dE/E0| < 1e-5 over 50 and "
                    "over 1 000 steps at N = 65 536 (absolute: a CPU step takes ~9 s there), and within 1e-5 of the CPU oracle "
                    "trajectory's drift at N = 8 192)",
    ("bf", "strict"): "strict math: bit-exact vs the oracle (tests/test_bf_gpu.py, tests/test_golden.py)",
    ("bh", "fast"): "fast math, a tolerance and (device tree) a distribution, not bit-exactness.  Host-built tree: node counts equal the oracle's, "
                    "every body within 1e-4 of its OWN |a| (median 1e-7, 99.9th percentile 1e-6 at N = 65 536; max 4.8e-6; 3.7e-5 at 2^20).  "
                    "Device-built tree (what NBODY_TREE_AUTO picks for fast math): same cells, centres of mass to the reference fold's own rounding, "
                    "so an opening test on its threshold can flip: accepted nodes within 1e-6 of the oracle's count, same median, 0 bodies beyond "
                    "1e-5 of their own |a| at N = 65 536 (max 8.9e-6), 217 of 2^20 beyond 1e-5 (max 4.9e-4 of own |a|, 2e-4 of max|a|) "
                    "(tests/test_bh_parity_evidence_gpu.py, both leaf rules; tests/test_bh_gpu.py, tests/test_bh_device_tree_gpu.py)",
    ("bh", "strict"): "strict math: accelerations and trajectories bit-exact vs the oracle (tests/test_bh_gpu.py, tests/test_large_gpu.py)",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["bf", "bh"], default="bf",
                    help="bf = configs[1] (65 536-body brute force, the metric's config, + the bh object on one GPU); "
                         "bh = configs[2] alone as the top-level line")
    ap.add_argument("--n", type=int, default=65536, help="bodies over all GPUs")
    ap.add_argument("--math", choices=["fast", "strict"], default="fast")
    ap.add_argument("--theta", type=float, default=0.5, help="Barnes-Hut opening angle (theta2 = theta^2)")
    ap.add_argument("--tree", choices=["host", "device"], default="host",
                    help="--workload bh: octree build on the host (north_star, bit-exact) or on the device (SURVEY F3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bh", action="store_true", help="default line without the configs[2] object")
    ap.add_argument("--profile-every", type=int, default=4, help="HIP events around every k-th launch of the dominant kernel (an event pair "
                    "costs the stream ~11 us: bracketing every launch makes the steps it measures 1.5-3 %% longer)")
    ap.add_argument("--spatial-n", type=int, default=1 << 22, help="N > 1 GPUs: bodies of the configs[4] object (0: leave it out)")
    ap.add_argument("--spatial-steps", type=int, default=10)
    ap.add_argument("--spatial-timeout", type=float, default=300.0, help="seconds before the configs[3]/configs[4] objects are given up")
    ap.add_argument("--large-n", type=int, default=1 << 20, help="N > 1 GPUs: bodies of the configs[3] object (0: leave it out)")
    ap.add_argument("--large-steps", type=int, default=5)
    ap.add_argument("--driver-n", type=int, default=100000, help="one GPU: disc bodies of the reference driver's workload (0: leave it out)")
    ap.add_argument("--main-timeout", type=float, default=600.0, help="N > 1 GPUs: seconds before the metric's own run is declared stuck")
    ap.add_argument("--preheat", type=int, default=-1,
                    help="untimed steps before the warm-up, reported as preheat_steps: the chip settles on the clock it holds under this kernel "
                         "only after ~50 steps (DESIGN.md 3.2), and the driver's invocation is 5 + 20 steps; -1: 100 at the metric's size, else 0")
    ap.add_argument("--seed", type=int, default=20250523)
    return ap.parse_args()


def load_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except Exception:
        return None


def cpu_baseline(orc, ics, settings, box, workload):
    """The oracle (a port of the reference's CPU path) on this box's host cores, bounded to ~10-30 s."""
    center, width = box
    n = len(ics)
    flags = orc.use_native()
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
            "value": m * (m - 1) / dt, "unit": "interactions/s", "cores": 1, "kind": "port", "build": flags,
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
        "value": acc / dt, "unit": "interactions/s", "cores": threads, "kind": "port", "build": flags,
        "steps_per_sec": 1.0 / dt,
        "sample": f"{reps} update_forces passes (recursive build + threaded recursive walk as "
                  f"barnes_hut.rs:143-203,250-263) over all {len(a)} bodies, {dt:.2f} s each",
    }


def bf_roofline(args, world, n, kernel_ms, launches, k_inter):
    """Dominant all-pairs kernel: fp32-VALU bound; the HBM fraction BASELINE asks for as a sub-field."""
    avg_ms = kernel_ms / max(1.0, launches)
    cross = os.environ.get("NBODY_CROSS_SYM", "1") != "0"
    kernel = ("k_bf_strict" if args.math == "strict" else
              "k_bf_sym" if (world == 1 and n >= 8192) else
              ("k_bf_cross" if cross else "k_bf_os") if (world > 1 and -(-n // world) >= 2048) else "k_bf_fast")
    inter_per_launch = k_inter / max(1.0, launches)
    tf = FLOP_PER_INTERACTION * inter_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    alg_bytes = BF_BYTES_PER_BODY * n / world
    gbs = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None
    if world == 1 and n == 65536 and args.math == "fast":
        pj = load_json("pmc_traffic_bf.json")
        traffic = pj.get("hbm_bytes_per_launch") if pj else None
    r = {
        "bound": "fp32-valu", "kernel": kernel, "achieved": tf, "peak": FP32_VALU_PEAK_TF, "unit": "TFLOP/s",
        "frac": tf / FP32_VALU_PEAK_TF, "traffic": traffic,
        "flop_per_interaction": FLOP_PER_INTERACTION, "interactions_per_launch": inter_per_launch,
        "avg_kernel_ms": avg_ms, "launches_timed": int(launches),
        # O(N) bytes for O(N^2) flops: the HBM fraction is tiny by construction
        "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": alg_bytes,
                "traffic_frac_of_peak": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and avg_ms > 0) else None},
    }
    if kernel in ("k_bf_sym", "k_bf_cross"):
        # the symmetric kernels evaluate every unordered pair once: 17 VALU ops = 26 flop per pair = 13 per
        # directed interaction actually issued; the 20-flop figure credits the one-sided formulation's work
        r["executed"] = {"flop_per_interaction": FLOP_EXECUTED_SYM, "achieved": tf * FLOP_EXECUTED_SYM / FLOP_PER_INTERACTION,
                         "frac": tf * FLOP_EXECUTED_SYM / FLOP_PER_INTERACTION / FP32_VALU_PEAK_TF, "unit": "TFLOP/s"}
    return r


def bh_roofline(n_local, kernel_ms, launches, visits_per_launch, tree, math="fast"):
    """The walk: L1/TA bound.  achieved = L1 cache-line accesses per cycle per CU = this run's
    opening tests (one 32-byte node record each, two 16-byte gathers) per cycle per CU, from its visit count and
    HIP-event kernel time, x the L1 line accesses per visit of the PMC pass kept in profiles/ (TCP_TOTAL_CACHE_ACCESSES
    and the visit count of that same pass); peak = the line rate the same access pattern sustains in isolation with
    fully divergent lanes (tools/microbench_gather.hip).  The visit rate against the microbenchmark's rows for 1 and
    2 lanes per record, and the HBM figure from the PMC bytes, ride along."""
    avg_ms = kernel_ms / max(1.0, launches)
    ceil = load_json("r02_gather_ceiling.json") or {}
    peak_lines = ceil.get("divergent_l1_line_accesses_per_cycle_per_cu")
    vpc = visits_per_launch / (avg_ms * 1e-3 * CLOCK_HZ * N_CU) if avg_ms > 0 else 0.0
    pj = load_json(f"pmc_traffic_bh_{tree}.json") or load_json("pmc_traffic_bh.json") or {}
    traffic = pj.get("hbm_bytes_per_launch")
    per_visit = None
    if pj.get("TCP_TOTAL_CACHE_ACCESSES_per_launch"):
        per_visit = pj["TCP_TOTAL_CACHE_ACCESSES_per_launch"] / pj.get("node_visits_per_launch", 1.2e8)  # both from the same PMC run
    lines = per_visit * vpc if per_visit else None
    rows = ceil.get("by_lanes_per_record_visits_per_cycle_per_cu") or {}
    return {
        "bound": "l1-ta", "kernel": ((pj.get("kernel") or "k_bh_walk") + " (+ k_bh_reduce)") if math == "fast" else "k_bh_walk_nested (strict: the reference's nested sums, one segment; the L1 figures below are the fast walk's)", "achieved": lines, "peak": peak_lines,
        "unit": "L1 cache-line accesses/cycle/CU", "frac": (lines / peak_lines) if (lines and peak_lines) else None,
        "traffic": traffic,
        "note": "since round 3 a lane walks several neighbouring bodies and fetches the union of their node sequences once: "
                "l1_line_accesses_per_visit < 1 is that sharing, and visits_per_cycle_per_cu may exceed the one-lane-per-record "
                "ceiling; `frac` is the L1 address rate actually used (what bounded the one-body walk), not a quality score of "
                "the faster kernel (DESIGN.md section 3.4)",
        "avg_kernel_ms": avg_ms, "launches_timed": int(launches), "visits_per_launch": visits_per_launch,
        "clock_hz_assumed": CLOCK_HZ, "l1_line_accesses_per_visit": per_visit,
        "visits_per_cycle_per_cu": vpc,
        "visits_ceiling_divergent": rows.get("1"), "visits_ceiling_2_lanes_per_record": rows.get("2"),
        "peak_source": ceil.get("source"),
        "hbm": {"traffic_frac_of_peak": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and avg_ms > 0) else None,
                "algorithmic_bytes_per_launch": BH_BYTES_PER_BODY * n_local, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "note": "node records are served by L1/L2; algorithmic HBM bytes are positions in + accelerations out"},
    }


_PARKED = []


def retire(sim, world):
    """A measured simulation is done with.  One rank: closed at once.  Several ranks: its communicator stays up until the
    line is out (main() closes the parked ones last): tearing a communicator down unmaps memory the peers exported, and on
    this driver an unmap under queues that are busy again a moment later -- the next object's steps -- froze every process
    of the device for tens of seconds with the one-device transport (profiles/r03_ipc_close_stall.txt).  Nothing the
    bench measures may depend on that, whatever the transport."""
    if world > 1:
        _PARKED.append(sim)
    else:
        sim.close()


def run(nb, args, workload, tree, ics, box, st, rank, world, local_rank, rdzv, ident_fn, preheat=0):
    """[preheat +] W warm-up + K timed steps of one workload; returns (elapsed, stats, n_after)."""
    n = len(ics)
    method = nb.BRUTE_FORCE if workload == "bf" else nb.BARNES_HUT
    math_mode = nb.FAST if args.math == "fast" else nb.STRICT
    sim = nb.Simulation(ics, *box, method=method, math_mode=math_mode, capacity=n, device=local_rank,
                        rank=rank, world_size=world,
                        tree_build=nb.TREE_DEVICE if tree == "device" else nb.TREE_HOST)
    sim.settings = nb.Settings(**st)
    if rdzv is not None and (world > 1 or os.environ.get("NBODY_BENCH_FORCE_COMM")):
        sim.comm_init(ident_fn())
    sim.init()

    def barrier():
        sim.sync()
        if rdzv is not None:
            rdzv.barrier()
            sim.sync()

    if preheat > 0:
        sim.steps(preheat)
    sim.steps(args.warmup)
    sim.set_profiling(max(1, args.profile_every))
    sim.reset_stats()
    barrier()
    t0 = time.perf_counter()
    sim.steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    stats = sim.stats()
    sim.set_profiling(False)
    n_after = sim.count_global() if world == 1 else None
    retire(sim, world)
    return elapsed, stats, n_after


def run_spatial(nb, args, box, st, rank, world, local_rank, rdzv, ident_fn):
    """configs[4]: Barnes-Hut over spatial shards with the halo exchange of nodes (NBODY_SHARD_SPATIAL, nbody_let.cpp),
    one rank per GPU over RCCL.  Returns this rank's record."""
    n = args.spatial_n
    ics = nb.plummer(n, seed=args.seed)
    sim = nb.Simulation(ics, *box, method=nb.BARNES_HUT, math_mode=nb.FAST, capacity=n, device=local_rank, rank=rank, world_size=world,
                        shard_mode=nb.SHARD_SPATIAL)
    sim.settings = nb.Settings(**st)
    sim.comm_init(ident_fn())
    sim.init()
    sim.steps(2)
    sim.set_profiling(max(1, args.profile_every))
    sim.reset_stats()
    sim.sync()
    rdzv.barrier()
    t0 = time.perf_counter()
    sim.steps(args.spatial_steps)
    sim.sync()
    rdzv.barrier()
    elapsed = time.perf_counter() - t0
    stats, ls = sim.stats(), sim.let_stats()
    own = len(sim)
    retire(sim, world)
    k = float(max(1, ls.steps))
    return {"elapsed": elapsed, "interactions": float(stats.interactions), "visits": float(stats.node_visits), "tree_nodes": int(stats.tree_nodes),
            "bodies": own, "nodes_local": ls.nodes_local / k, "nodes_sent": ls.nodes_sent / k, "nodes_received": ls.nodes_received / k,
            "bytes_sent": ls.bytes_sent / k, "bytes_allgather": ls.bytes_allgather_equivalent / k, "migrated": ls.bodies_migrated / k,
            "phase_ms": [ls.phase_ms[i] / k for i in range(5)]}


def bh_record(args, n, tree, elapsed, stats):
    visits = float(stats.node_visits)
    launches = float(stats.force_launches)
    rec = {
        "workload": f"configs[2]: {n}-body Barnes-Hut theta={args.theta}" if n == 65536 else f"bh n={n}",
        "tree_build": tree, "math": args.math,
        "ms_per_step": 1e3 * elapsed / args.steps, "steps_per_sec": args.steps / elapsed,
        "interactions_per_sec": float(stats.interactions) / elapsed,
        "tree_nodes": int(stats.tree_nodes),
        "node_visits_per_step": visits / args.steps, "accepted_per_step": float(stats.interactions) / args.steps,
        "split_ms_per_step": {
            "tree_build": stats.tree_build_ms / args.steps,     # host wall time in the build (device build: enqueue + read-back wait)
            "tree_copy": stats.tree_copy_ms / args.steps,       # host build only: D2H positions + H2D nodes (incl. waiting for the previous step)
            "walk_kernel": stats.force_kernel_ms / max(1.0, launches),
        },
        "roofline": bh_roofline(n, stats.force_kernel_ms, launches, visits / max(1, args.steps), tree, args.math),   # (one walk launch per step)
        "parity": PARITY[("bh", args.math)],
    }
    return rec


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if os.environ.get("NBODY_BENCH_DEVICE"):   # rehearsal on a box with fewer GPUs than ranks (if RCCL accepts it)
        local_rank = int(os.environ["NBODY_BENCH_DEVICE"])
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched through torch.distributed.run (one rank per GPU)")
        args.gpus = world

    rdzv = None
    json_fd = None
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # by torch.distributed.run (any N)
    if launched:
        # RCCL prints connection/version banners on stdout: stdout is pointed at stderr for the whole run and the one JSON
        # line goes to the original descriptor at the end
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)

    def emit(line_obj):
        data = (json.dumps(line_obj) + "\n").encode()
        if json_fd is not None:
            os.write(json_fd, data)
        else:
            sys.stdout.write(data.decode())
            sys.stdout.flush()

    def error_line(text):
        return {"metric": "pairwise_interactions_per_sec", "value": None, "unit": "interactions/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "higher_is_better": True, "error": text}

    nb = graft.load_package()
    if nb.device_count() < 1:
        sys.exit("bench.py needs a HIP device: the engine has no CPU fallback")
    if launched:
        from nbody_llm_amd.rendezvous import Rendezvous   # stdlib control plane: no torch in a rank process
        rdzv = Rendezvous(rank, world, timeout=float(os.environ.get("NBODY_BENCH_RDZV_TIMEOUT", "300")))

    n = args.n
    box = ((0.0, 0.0, 0.0), 64.0)
    theta2 = args.theta * args.theta
    st = dict(g=1.0, g_soft=1e-2, dt=1e-3, theta2=theta2)
    ics = nb.plummer(n, seed=args.seed)
    preheat = args.preheat if args.preheat >= 0 else (100 if (args.workload == "bf" and n == 65536 and args.math == "fast") else 0)

    def ident_fn():
        # (NBODY_TRANSPORT=ipc in the environment makes this the id of the one-device transport: the one-GPU rehearsal)
        return rdzv.bcast_bytes(nb.comm_unique_id() if rank == 0 else None)

    main_guard = None
    if world > 1:
        # a rank that never comes back from a collective would leave the driver with nothing at all: after --main-timeout
        # rank 0 says so in the one line and every rank leaves
        import threading

        def stuck():
            if rank == 0:
                emit(error_line(f"no result within {args.main_timeout:.0f} s (a rank stuck in the exchange?)"))
            os._exit(3)

        main_guard = threading.Timer(args.main_timeout, stuck)
        main_guard.daemon = True
        main_guard.start()
    try:
        elapsed, stats, n_after = run(nb, args, args.workload, args.tree, ics, box, st, rank, world, local_rank, rdzv, ident_fn, preheat=preheat)
        mine = [elapsed, float(stats.interactions), stats.force_kernel_ms, float(stats.force_launches), float(stats.node_visits),
                float(stats.force_kernel_interactions)]
        gathered = rdzv.allgather(mine) if rdzv is not None else [mine]
    except BaseException as e:  # noqa: BLE001 -- a rank that raises must not leave the driver without a line, nor its peers in a collective
        if rank == 0:
            emit(error_line(f"rank 0: {type(e).__name__}: {e}"))
        else:
            sys.stderr.write(f"bench.py rank {rank}: {type(e).__name__}: {e}\n")
        os._exit(1)   # (closes this rank's sockets: a peer blocked in the rendezvous sees the connection go and lands here too)
    elapsed = max(float(g[0]) for g in gathered)
    interactions = sum(float(g[1]) for g in gathered)
    kernel_ms = max(float(g[2]) for g in gathered)   # the slowest rank's kernel time
    launches = float(gathered[0][3])
    visits = sum(float(g[4]) for g in gathered)
    k_inter = float(gathered[0][5])                  # rank 0's dominant-kernel interactions

    if main_guard is not None:
        main_guard.cancel()
    result = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = interactions / elapsed
        if args.workload == "bf":
            roofline = bf_roofline(args, world, n, kernel_ms, launches, k_inter)
        else:
            roofline = bh_roofline(n / world, kernel_ms, launches, (visits / world) / max(1, args.steps), args.tree, args.math)
        result = {
            "metric": "pairwise_interactions_per_sec", "value": value, "unit": "interactions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "steps_per_sec": args.steps / elapsed,
            "preheat_steps": preheat,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": ("configs[1]: 65 536-body brute force" if (args.workload == "bf" and n == 65536) else
                             "configs[2]: 65 536-body Barnes-Hut theta=0.5" if (args.workload == "bh" and n == 65536) else
                             f"{args.workload} n={n}"),
                "n_bodies": n, "method": "brute_force" if args.workload == "bf" else "barnes_hut",
                "math": args.math, "ics": f"plummer seed={args.seed}", "dt": st["dt"], "g_soft": st["g_soft"],
                "theta2": theta2 if args.workload == "bh" else None, "box_width": box[1],
                "parallelism": "1 GPU" if world == 1 else
                               f"{world} index-block shards; per step one all-gather of positions and one "
                               f"send/recv round of partial sums (every pair between shards evaluated once); transport "
                               f"{'ipc (all ranks on one device: rehearsal)' if os.environ.get('NBODY_TRANSPORT') == 'ipc' else 'rccl'}",
                "bodies_left_in_box": n_after,
            },
            "roofline": roofline,
            "parity": PARITY[(args.workload, args.math)],
        }
        if args.workload == "bh":
            result["bh"] = bh_record(args, n, args.tree, elapsed, stats)

    # configs[2] beside the metric's line: both tree builds, one GPU only (the driver's N = 1 record)
    if args.workload == "bf" and world == 1 and not args.no_bh:
        bh = {}
        for tree in ("host", "device"):
            e2, s2, _ = run(nb, args, "bh", tree, ics, box, st, rank, world, local_rank, None, None)
            bh[f"{tree}_tree"] = bh_record(args, n, tree, e2, s2)
        if rank == 0:
            result["bh"] = bh

    # configs[4] beside the metric's line when there is more than one GPU: Barnes-Hut over spatial shards.  Its RCCL
    # exchange runs with > 1 rank only here (a one-GPU box cannot rehearse it), so it is fenced: an error or a rank that
    # does not come back within --spatial-timeout leaves a note in the line instead of taking the metric with it.
    extras = world > 1 or (rdzv is not None and os.environ.get("NBODY_BENCH_FORCE_EXTRAS"))   # (the env switch: a 1-rank rehearsal of this block)
    if args.workload == "bf" and extras and not args.no_bh and (args.spatial_n > 0 or args.large_n > 0):
        import threading

        def give_up():
            if rank == 0:
                result.setdefault("bf_large", {"error": f"no result within {args.spatial_timeout:.0f} s"})
                result["bh_spatial"] = {"error": f"no result within {args.spatial_timeout:.0f} s"}
                result["cpu_baseline"] = None
                emit(result)
            os._exit(0)

        guard = threading.Timer(args.spatial_timeout, give_up)
        guard.daemon = True
        guard.start()
        # configs[3]: brute force at 2^20 bodies over the same index-block shards as the metric's line
        big = None
        if args.large_n > 0:
            try:
                import copy
                a3 = copy.copy(args)
                a3.steps, a3.warmup = args.large_steps, 1
                e3, s3, _ = run(nb, a3, "bf", args.tree, nb.plummer(args.large_n, seed=args.seed), box, st, rank, world, local_rank, rdzv, ident_fn)
                big = {"elapsed": e3, "interactions": float(s3.interactions), "kernel_ms": s3.force_kernel_ms / max(1.0, float(s3.force_launches))}
            except Exception as e:  # noqa: BLE001
                big = {"error": f"{type(e).__name__}: {e}"}
            bigs = rdzv.allgather(big)
            if rank == 0:
                bad = [b for b in bigs if b is None or "error" in b]
                if bad:
                    result["bf_large"] = {"error": (bad[0] or {}).get("error", "a rank returned nothing")}
                else:
                    el = max(b["elapsed"] for b in bigs)
                    result["bf_large"] = {
                        "workload": f"configs[3]: {args.large_n}-body brute force, {world} index-block shards, all-gather of positions + "
                                    f"one send/recv round of partial sums per step (transport {'ipc' if os.environ.get('NBODY_TRANSPORT') == 'ipc' else 'rccl'})",
                        "ms_per_step": 1e3 * el / args.large_steps, "steps_per_sec": args.large_steps / el,
                        "interactions_per_sec": sum(b["interactions"] for b in bigs) / el, "steps": args.large_steps,
                        "cross_kernel_ms_per_rank": [b["kernel_ms"] for b in bigs],
                        "parity": PARITY[("bf", args.math)],
                    }
        rec = None
        try:
            if args.spatial_n > 0:
                rec = run_spatial(nb, args, box, dict(st, theta2=theta2), rank, world, local_rank, rdzv, ident_fn)
            else:
                rec = {"skipped": True}
        except Exception as e:  # noqa: BLE001 -- whatever it is, the metric's line must still go out
            rec = {"error": f"{type(e).__name__}: {e}"}
        recs = rdzv.allgather(rec)
        guard.cancel()
        if rank == 0:
            bad = [r for r in recs if r is None or "error" in r]
            if all(r and r.get("skipped") for r in recs):
                pass
            elif bad:
                result["bh_spatial"] = {"error": (bad[0] or {}).get("error", "a rank returned nothing")}
            else:
                el = max(r["elapsed"] for r in recs)
                k = args.spatial_steps
                result["bh_spatial"] = {
                    "workload": f"configs[4]: {args.spatial_n}-body Barnes-Hut theta={args.theta}, {world} spatial shards (Morton-key ranges), "
                                f"halo exchange of tree nodes (transport {'ipc' if os.environ.get('NBODY_TRANSPORT') == 'ipc' else 'rccl'})",
                    "ms_per_step": 1e3 * el / k, "steps_per_sec": k / el, "interactions_per_sec": sum(r["interactions"] for r in recs) / el,
                    "tree_nodes": recs[0]["tree_nodes"], "steps": k,
                    "per_rank": [{kk: r[kk] for kk in ("bodies", "nodes_local", "nodes_sent", "nodes_received", "bytes_sent", "bytes_allgather",
                                                       "migrated", "phase_ms")} for r in recs],
                    "phase_names": ["drift+retain+pick migrants", "append+keys+sort", "scans+spanning-cell table", "emit+flag+pack export",
                                    "walk+kick"],
                    "note": "phase_ms = device time of each rank's kernels between the exchanges (HIP events); the exchanges themselves are in ms_per_step only",
                    "parity": "fast math, device build: node counts equal the single-GPU tree's, accelerations to rounding (tests/test_spatial_gpu.py, "
                              "tests/test_large_gpu.py at this size, one-GPU emulation of the ranks)",
                }

    # the reference's own benchmark (src/main.rs:52-129 through its Simulation trait): the 100 000-body disc, Barnes-Hut,
    # theta2 = 1, dt = 3e-2, g_soft = 0.02 -- the only workload with published numbers (its authors' perf CSVs,
    # other hardware, f64, 32 CPU threads: context, not a vs_baseline)
    if args.workload == "bf" and world == 1 and not args.no_bh and args.driver_n > 0:
        ref = {}
        rbox = ((0.0, 0.0, 0.0), 10.0)
        rst = dict(g=1.0, g_soft=0.02, dt=3e-2, theta2=1.0)
        for name, f64, tree, mm, k in (("f32_fast_device_tree", False, nb.TREE_DEVICE, nb.FAST, 300), ("f32_strict_host_tree", False, nb.TREE_HOST, nb.STRICT, 60),
                                       ("f64_host_tree", True, nb.TREE_HOST, nb.STRICT, 40), ("f64_device_tree", True, nb.TREE_DEVICE, nb.STRICT, 120),
                                       ("f64_fast_device_tree", True, nb.TREE_DEVICE, nb.FAST, 200)):
            d = nb.disc(args.driver_n, seed=1, f64=f64)
            with nb.Simulation(d, *rbox, method=nb.BARNES_HUT, math_mode=mm, tree_build=tree, f64=f64) as sim:
                sim.settings = nb.Settings(**rst)
                sim.init()
                sim.steps(5)
                sim.sync()
                t0 = time.perf_counter()
                sim.steps(k)
                sim.sync()
                ref[name] = {"steps_per_sec": k / (time.perf_counter() - t0), "steps": k, "bodies_left": len(sim)}
        if rank == 0:
            result["reference_driver"] = {
                "workload": f"src/main.rs: 1 star + {args.driver_n} disc bodies, box 10, Barnes-Hut theta2 = 1.0, dt = 3e-2, g_soft = 0.02 "
                            "(what `nbody_cli -t T -n N` and the reference's perf_benchmark.py run)",
                **ref,
                "published_cpu": {"steps_per_sec": 8.1, "what": "the reference on a 32-thread AMD Zen host, f64, N = 100 000 (BASELINE.md, "
                                                                "combined_nbody_man_opt.csv:2881): other hardware, context only"},
                "parity": "f64_host_tree and f32_strict_host_tree: bit-exact vs the oracle (tests/test_f64_gpu.py::test_reference_driver_configuration_in_f64, "
                          "tests/test_bh_gpu.py); the device-tree entries: same cells, node counts within 1e-3 (f32) / 1e-6 (f64); f64_fast_device_tree: one "
                          "running sum per lane over a split node range, accelerations to 1e-12 of the oracle's (tests/test_f64_gpu.py)",
            }

    if rank == 0:
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is timed at N = 1 only
            orc = graft.load_oracle()
            result["cpu_baseline"] = cpu_baseline(orc, ics, st, box, args.workload)
            if "bh" in result and args.workload == "bf":
                result["bh"]["cpu_baseline"] = cpu_baseline(orc, ics, st, box, "bh")
        else:
            result["cpu_baseline"] = None
        emit(result)
    if rdzv is not None:
        rdzv.barrier()
    for sim in _PARKED:   # (the line is out: see retire())
        try:
            sim.close()
        except Exception:  # noqa: BLE001 -- nothing depends on it any more
            pass
    if rdzv is not None:
        rdzv.close()


if __name__ == "__main__":
    main()
