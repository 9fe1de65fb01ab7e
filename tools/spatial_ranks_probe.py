"""Wall time per step of the spatial-shard Barnes-Hut step, real ranks over the one-device IPC transport, by size and
rank count (a rehearsal of the control flow, not a scaling figure: the ranks share one GPU).
    python tools/spatial_ranks_probe.py 4:1048576 4:2097152 2:2097152"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

nb = graft.load_package()
from nbody_llm_amd import ranks  # noqa: E402

steps = int(os.environ.get("PROBE_STEPS", "6"))
for spec in sys.argv[1:]:
    g, n = (int(x) for x in spec.split(":"))
    with tempfile.TemporaryDirectory() as out:
        cfg = {"world": g, "out": out, "transport": "ipc", "device": 0,
               "sim": {"method": "bh", "math": "fast", "shard": "spatial", "tree": "device"},
               "ics": {"kind": "plummer", "n": n, "seed": 20250523}, "box": [[0, 0, 0], 400.0],
               "settings": {"g": 1.0, "g_soft": 0.05, "dt": 1e-3, "theta2": 0.25},
               "schedule": json.loads(os.environ["PROBE_SCHEDULE"]) if os.environ.get("PROBE_SCHEDULE") else [["steps", steps]], "env": json.loads(os.environ.get("PROBE_ENV", "{}"))}
        res = ranks.run_world(cfg, timeout=600)
        if cfg["env"].get("NBODY_LET_TRACE"):
            for g in range(cfg["world"]):
                print(open(os.path.join(out, f"proc{g}.log")).read())
        for r in res:
            print(spec, "rank", r["rank"], "bodies", r["count"], "wall/step ms", round(1e3 * r["wall_s"] / steps, 2), r["let"], flush=True)
