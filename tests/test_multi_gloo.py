"""world_size-2 rehearsal of the sharded brute-force step on CPU with torch.distributed/gloo.

What is exercised is the sharding SCHEME the HIP library implements (nbody_upload's contiguous
index blocks, one all-gather of half-drifted positions + live counts per step, per-shard retain)
and the control-plane pattern of a rank process (id from rank 0 to everybody, max-over-ranks; bench.py's own
control plane, nbody-llm_amd/rendezvous.py, is covered by tests/test_rendezvous.py): each
rank advances its own block with the oracle's row-wise force over the gathered positions, and the
concatenation must equal the unsharded oracle bit for bit.  (The device kernels themselves are
covered by tests/test_sharded_gpu.py.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, steps, out_dir):
    sys.path.insert(0, ROOT)
    import torch                       # BEFORE the HIP library, like bench.py with N > 1: one process
    import torch.distributed as dist   # must not hold two ROCm runtimes (torch bundles its own)
    import __graft_entry__ as graft
    nb = graft.load_package()
    orc = graft.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    # control plane as in bench.py: rank 0's 128-byte id reaches every rank
    ident = [bytes(range(128)) if rank == 0 else None]
    dist.broadcast_object_list(ident, src=0)
    assert ident[0] == bytes(range(128))

    box = ((0.0, 0.0, 0.0), 1.6)
    sd = dict(g=1.0, g_soft=0.05, dt=2e-2, theta2=0.5)
    ics = nb.plummer(n, seed=99)       # every rank generates the same bodies
    lo, hi = nb.shard_range(n, rank, world)
    cap = -(-n // world)
    own = ics[lo:hi].copy().astype(orc.P32)
    for _ in range(steps):
        orc.pre_force(own, sd["dt"])                               # K1
        own = orc.retain(own, box[0], box[1])                       # K4 (per shard, order kept)
        # exchange: fixed-size padded blocks + live counts (what ncclAllGather moves)
        blk = np.zeros((cap, 4), np.float32)
        blk[: len(own), :3] = own["position"]
        blk[: len(own), 3] = own["mass"]
        send = torch.from_numpy(blk)
        recv = [torch.zeros_like(send) for _ in range(world)]
        dist.all_gather(recv, send)
        cnt = [torch.zeros(1, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(cnt, torch.tensor([len(own)], dtype=torch.int32))
        counts = [int(c[0]) for c in cnt]
        # forces of the own block against all live bodies, partners in ascending global order
        allb = np.zeros(sum(counts), dtype=orc.P32)
        at, first = 0, None
        for r in range(world):
            seg = recv[r].numpy()[: counts[r]]
            if r == rank:
                first = at
            allb["position"][at: at + counts[r]] = seg[:, :3]
            allb["mass"][at: at + counts[r]] = seg[:, 3]
            at += counts[r]
        orc.bf_update_forces_range(allb, sd, first, first + len(own), threads=1)
        own["acceleration"] = allb["acceleration"][first: first + len(own)]
        orc.after_force(own, sd["dt"])                              # K3
    # the send/recv round of the symmetric scheme across shards, with the plan the library uses
    # (nbody_host_cross_plan): one message to every shard this rank is resident for, one from every
    # shard that is resident for it -- must complete (no deadlock) and deliver the right senders' data
    plan = nb.host_cross_plan(rank, world, cap, hi - lo)
    ops, bufs = [], []
    for seg in plan["parts"][:, 0]:
        ops.append(dist.P2POp(dist.isend, torch.full((cap * 4,), float(rank * 100 + int(seg))), int(seg)))
    for src in plan["recv_from"]:
        b = torch.zeros(cap * 4)
        bufs.append((int(src), b))
        ops.append(dist.P2POp(dist.irecv, b, int(src)))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for src, b in bufs:
        assert torch.all(b == float(src * 100 + rank)), (rank, src)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), own)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)        # max-over-ranks like bench.py
    g = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(g, t)
    assert max(float(x[0]) for x in g) == float(world)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_scheme_equals_unsharded_oracle(tmp_path, nb, orc, world):
    # ranks are plain child processes (this pytest process never imports torch: see conftest.nb)
    import subprocess
    n, steps = 700, 6
    port = 29500 + (os.getpid() % 2000) + world
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), str(world), str(port), str(n),
                               str(steps), str(tmp_path)]) for r in range(world)]
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0] * world
    got = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    box = ((0.0, 0.0, 0.0), 1.6)
    sd = dict(g=1.0, g_soft=0.05, dt=2e-2, theta2=0.5)
    ref = nb.plummer(n, seed=99).astype(orc.P32)
    for _ in range(steps):
        ref = orc.bf_step_by(ref, sd, box[0], box[1], sd["dt"])
    assert len(ref) < n, "the case must drop bodies"
    assert len(got) == len(ref)
    for f in ("position", "velocity", "acceleration", "mass"):
        assert np.array_equal(got[f], ref[f]), f


if __name__ == "__main__":
    _worker(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6])
