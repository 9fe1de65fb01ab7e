"""The stdlib control plane of a multi-rank run (nbody-llm_amd/rendezvous.py): what bench.py and the rank worker use
instead of torch.distributed.  CPU only: ranks as threads and as fresh processes."""
import os
import subprocess
import sys
import tempfile
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _addr(tag):
    return os.path.join(tempfile.gettempdir(), f"nbody_rdzv_test_{os.getpid()}_{tag}.sock")


def test_collectives_between_threads(nb):
    from nbody_llm_amd.rendezvous import Rendezvous
    world, addr, out = 5, _addr("t"), {}

    def rank_main(r):
        rz = Rendezvous(r, world, addr, timeout=30)
        ident = rz.bcast_bytes(bytes(range(128)) if r == 0 else None)
        g = rz.gather({"rank": r, "x": r * r})
        a = rz.allgather(r + 100)
        rz.barrier()
        out[r] = (ident, g, a)
        rz.close()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in reversed(ts):      # rank 0 arrives last: the others retry until it listens
        t.start()
    for t in ts:
        t.join(60)
    assert sorted(out) == list(range(world))
    for r in range(world):
        ident, g, a = out[r]
        assert ident == bytes(range(128))
        assert a == [100 + k for k in range(world)]
        assert g == ([{"rank": k, "x": k * k} for k in range(world)] if r == 0 else None)
    assert not os.path.exists(addr)


def test_world_of_one_needs_no_socket(nb):
    from nbody_llm_amd.rendezvous import Rendezvous
    rz = Rendezvous(0, 1, _addr("one"))
    assert rz.bcast_bytes(b"abc") == b"abc" and rz.gather(7) == [7] and rz.allgather(1) == [1]
    rz.barrier()
    rz.close()


def test_ranks_as_processes_find_each_other_through_master_port(nb, tmp_path):
    """what bench.py does under `python -m torch.distributed.run`: the address comes from MASTER_PORT in the environment"""
    code = (
        "import os, sys, json\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import __graft_entry__ as g\n"
        "g.load_package()\n"
        "from nbody_llm_amd.rendezvous import Rendezvous\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "rz = Rendezvous(r, w, timeout=60)\n"
        "got = rz.allgather({'rank': r, 'pid': os.getpid()})\n"
        "rz.barrier(); rz.close()\n"
        "print(json.dumps(got))\n")
    env = dict(os.environ, MASTER_PORT=str(20000 + os.getpid() % 20000), WORLD_SIZE="3")
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, text=True) for r in range(3)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    import json
    views = [json.loads(o.strip().splitlines()[-1]) for o in outs]
    assert views[0] == views[1] == views[2] and [v["rank"] for v in views[0]] == [0, 1, 2]
    assert len({v["pid"] for v in views[0]}) == 3
