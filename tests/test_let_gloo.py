"""world_size-2/3/4 rehearsal on CPU (torch.distributed/gloo) of the exchanges the spatial-shard Barnes-Hut step makes
(nbody_let.cpp: all-gather of the ranks' boxes, all-gather of the spanning cells' partial sums, counts matrix + ONE
variable-size send/recv round of node records, scatter by global index), with the export rule of kernels_let.hip restated
in numpy: a node goes to a partner iff a body inside the partner's bounding box could OPEN every one of its ancestors.
The tree is the oracle's (every rank builds it from the same bodies and then FORGETS every node it does not own), so what
is proven is the protocol and the rule: after the exchange each rank walks its own bodies over its assembled array without
ever touching a node it does not hold, and gets the forces of the complete tree bit for bit.  (The device kernels are
covered by tests/test_spatial_gpu.py on one GPU; RCCL itself needs one GPU per rank.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOX = ((0.0, 0.0, 0.0), 8.0)
SD = dict(g=1.0, g_soft=0.05, dt=1e-3, theta2=0.64)


def tree_links(tree):
    """parent of every node, and for every node the first / last LEAF of its subtree (pre-order array with skip links)"""
    skip, nchild = tree["skip"], tree["nchild"]
    m = len(skip)
    parent = np.full(m, -1, np.int64)
    stack = []
    for i in range(m):
        while stack and skip[stack[-1]] <= i:
            stack.pop()
        parent[i] = stack[-1] if stack else -1
        if nchild[i] > 0:
            stack.append(i)
    leaf = nchild == 0
    leaf_no = np.cumsum(leaf) - 1                      # number of the last leaf at or before i
    first_leaf = np.where(leaf, leaf_no, leaf_no + 1)  # pre-order: the subtree of i starts at i
    last_leaf = leaf_no[skip - 1]
    return parent, leaf, first_leaf, last_leaf


def could_open(com, w2, lo, hi, theta2):
    """kernels_let.hip box_could_open: can a point of the box fail w^2 < theta2 r^2 on this node (f32, margin 0.9999)"""
    d = np.maximum(np.float32(0), np.maximum(lo - com, com - hi)).astype(np.float32)
    d2 = np.float32((d * d).sum(dtype=np.float32))
    return not (np.float32(w2) < np.float32(theta2) * d2 * np.float32(0.9999))


def walk(nodes, skip, leaf, p, sd, root_stop, visited):
    """barnes_hut.rs:185-203 over a pre-order array (f64 arithmetic: both walks compared here use the same code)"""
    acc = np.zeros(3)
    i = 0
    eps2 = sd["g_soft"] ** 2
    while i < root_stop:
        visited.append(i)
        rec = nodes[i]
        d = rec[:3].astype(np.float64) - p
        r2 = float(d @ d)
        if float(rec[4]) < sd["theta2"] * r2:
            acc += sd["g"] * float(rec[3]) * d / (r2 + eps2) ** 1.5
            i = skip[i]
        elif leaf[i]:
            i = skip[i]                                # the reference drops a leaf that fails the test
        else:
            i += 1
    return acc


def _worker(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    nb = graft.load_package()
    orc = graft.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    ics = nb.plummer(n, seed=77).astype(orc.P32)
    ics["position"] *= np.float32(0.5)
    tree = orc.bh_build_tree(ics, BOX[0], BOX[1])
    m = len(tree["skip"])
    parent, leaf, first_leaf, last_leaf = tree_links(tree)
    n_leaves = int(leaf.sum())
    assert n_leaves == n
    full = np.concatenate([tree["com_mass"], (tree["width"] ** 2)[:, None]], axis=1).astype(np.float32)   # {com, m, w^2}
    owner_of_leaf = np.minimum(world - 1, np.arange(n) * world // n)         # contiguous runs of the depth-first leaf order
    own_first, own_last = owner_of_leaf[first_leaf], owner_of_leaf[last_leaf]
    spanning = own_first != own_last
    mine = (own_first == rank) & ~spanning                                    # a cell belongs to the rank of its first body
    my_leaves = np.flatnonzero(leaf & (own_first == rank))
    my_bodies = tree["leaf_body"][my_leaves]

    # exchange 1 (fixed size, all-gather): every rank's bounding box
    pos = ics["position"][my_bodies]
    box = torch.from_numpy(np.concatenate([pos.min(0), pos.max(0)]).astype(np.float32))
    boxes = [torch.zeros(6) for _ in range(world)]
    dist.all_gather(boxes, box)
    boxes = [b.numpy() for b in boxes]

    # exchange 2 (fixed size, all-gather): what each rank adds to every spanning cell, summed in rank order
    span_ids = np.flatnonzero(spanning)
    part = np.zeros((len(span_ids), 4))
    for j, c in enumerate(span_ids):
        inside = my_leaves[(my_leaves > c) & (my_leaves < tree["skip"][c])]
        b = tree["leaf_body"][inside]
        mm = ics["mass"][b].astype(np.float64)
        part[j, 0] = mm.sum()
        part[j, 1:] = (mm[:, None] * ics["position"][b].astype(np.float64)).sum(0)
    parts = [torch.zeros(part.shape, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(part))
    tot = sum(p.numpy() for p in parts)
    assembled = np.full((m, 5), np.nan, np.float32)
    assembled[mine] = full[mine]
    assembled[span_ids, 3] = tot[:, 0]
    assembled[span_ids, :3] = tot[:, 1:] / tot[:, :1]
    assembled[span_ids, 4] = full[span_ids, 4]
    assert np.allclose(assembled[span_ids, :4], full[span_ids, :4], rtol=1e-5, atol=1e-5)   # (against the f32 sequential folds of the reference build)
    assembled[span_ids] = full[span_ids]               # (the f32 folds of the reference's build: keep its bits for the comparison)

    # the export rule: partner p gets my node i iff p could open every ancestor of i
    open_by = np.zeros((m, world), bool)
    for i in np.flatnonzero(~leaf):
        for p in range(world):
            open_by[i, p] = could_open(full[i, :3], full[i, 4], boxes[p][:3], boxes[p][3:], SD["theta2"])
    need = np.ones((m, world), bool)
    for i in range(1, m):                              # pre-order: a parent comes before its children
        need[i] = need[parent[i]] & open_by[parent[i]]
    lists = {p: np.flatnonzero(mine & need[:, p]) for p in range(world) if p != rank}

    # exchange 3: the counts matrix (all-gather), then ONE round of variable-size messages
    row = torch.tensor([len(lists[p]) if p != rank else 0 for p in range(world)], dtype=torch.int64)
    rows = [torch.zeros(world, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(rows, row)
    matrix = torch.stack(rows).numpy()                 # matrix[r][p] = records r sends to p
    ops, inbox = [], {}
    for p in range(world):
        if p == rank:
            continue
        if matrix[rank][p] > 0:
            rec = np.concatenate([full[lists[p]], lists[p][:, None].astype(np.float32)], axis=1)   # record + global index
            ops.append(dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(rec)), p))
        if matrix[p][rank] > 0:
            inbox[p] = torch.zeros((int(matrix[p][rank]), 6))
            ops.append(dist.P2POp(dist.irecv, inbox[p], p))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for p, buf in inbox.items():
        rec = buf.numpy()
        idx = rec[:, 5].astype(np.int64)
        assert np.isnan(assembled[idx, 0]).all() and (own_first[idx] == p).all()   # nothing I hold is sent to me, every record from its owner
        assembled[idx] = rec[:, :5]

    # the walk never leaves what this rank holds, and gives the forces of the complete tree, bit for bit
    acc = np.zeros((len(my_bodies), 3))
    touched = 0
    for j, b in enumerate(my_bodies):
        v1, v2 = [], []
        a_full = walk(full, tree["skip"], leaf, ics["position"][b].astype(np.float64), SD, m, v1)
        a_mine = walk(assembled, tree["skip"], leaf, ics["position"][b].astype(np.float64), SD, m, v2)
        assert v1 == v2 and not np.isnan(assembled[v2, 0]).any(), (rank, int(b))
        assert np.array_equal(a_full, a_mine)
        acc[j] = a_mine
        touched += len(v2)
    held = int((~np.isnan(assembled[:, 0])).sum())
    np.savez(os.path.join(out_dir, f"let{rank}.npz"), bodies=my_bodies, acc=acc, held=held, nodes=m,
             sent=int(matrix[rank].sum()), received=int(matrix[:, rank].sum()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_let_exchange_protocol_and_export_rule(tmp_path, nb, orc, world):
    import subprocess
    n = 500
    port = 31500 + (os.getpid() % 2000) + world
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), str(world), str(port), str(n), str(tmp_path)])
             for r in range(world)]
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0] * world
    ics = nb.plummer(n, seed=77).astype(orc.P32)
    ics["position"] *= np.float32(0.5)
    ref = ics.copy()
    orc.bh_update_forces(ref, SD, BOX[0], BOX[1], threads=1)
    got = np.zeros((n, 3))
    seen = np.zeros(n, bool)
    held, sent = [], []
    for r in range(world):
        z = np.load(tmp_path / f"let{r}.npz")
        assert not seen[z["bodies"]].any()
        seen[z["bodies"]] = True
        got[z["bodies"]] = z["acc"]
        held.append(int(z["held"]))
        sent.append(int(z["sent"]))
        nodes = int(z["nodes"])
    assert seen.all()
    scale = np.abs(ref["acceleration"]).max()
    assert np.abs(got - ref["acceleration"]).max() / scale < 1e-5     # f64 walk of the same nodes against the oracle's f32 walk
    assert sum(held) < world * nodes                                  # the rule does prune (little at 500 bodies: the cells are big against the boxes)
    print(f"world {world}: {nodes} nodes, held per rank {held}, sent per rank {sent}")


if __name__ == "__main__":
    _worker(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
