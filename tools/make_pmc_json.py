"""Turns the outputs of tools/profile_all.sh (gpurun_out/prof_<tag>/) into the small JSON files bench.py reads:
profiles/pmc_traffic_bf.json, profiles/pmc_traffic_bh_host.json, profiles/pmc_traffic_bh_device.json.
HBM bytes per launch of the dominant kernel = 2 x FETCH_SIZE (gfx950 counts the 128-byte requests of 16-byte-per-lane
reads at 64 bytes, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both reported in KiB by rocprofv3.
    python tools/make_pmc_json.py r02 <commit>"""
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
commit = sys.argv[2] if len(sys.argv) > 2 else "?"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
text = open(os.path.join(OUT, "pmc_summary.txt")).read()


def counter(section, kernel_sub):
    """average per dispatch of the counter in `== ... : section` for the first kernel whose name contains kernel_sub"""
    m = re.search(rf"== rocprofv3 --pmc \S+: {section}\n(.*?)(?=\n== |\Z)", text, re.S)
    if not m:
        return None
    for line in m.group(1).splitlines():
        if kernel_sub in line:
            global last_kernel_name
            last_kernel_name = re.search(r"(%s\w*<[^>]*>)" % re.escape(kernel_sub.rstrip("<")), line).group(1)
            return float(line.split()[-1])
    return None


last_kernel_name = None


def bench(name):
    with open(os.path.join(OUT, f"{name}.bench.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def write(name, d):
    with open(os.path.join(ROOT, "profiles", name), "w") as f:
        json.dump(d, f, indent=1)
    print(name, json.dumps(d)[:300])


fz, wz = counter("pmc_fetch_bf", "k_bf_sym<"), counter("pmc_write_bf", "k_bf_sym<")
write("pmc_traffic_bf.json", {
    "kernel": last_kernel_name, "workload": "configs[1] N=65536 brute force, fast math, 1 GPU", "commit": commit,
    "FETCH_SIZE_KiB_per_launch": fz, "WRITE_SIZE_KiB_per_launch": wz,
    "correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B for 16-B/lane streaming reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE exact",
    "hbm_bytes_per_launch": int(2 * fz * 1024 + wz * 1024),
    "note": "the writes are the partial-sum planes that replace atomics (read back by k_bf_sym_reduce)",
    "source": f"tools/profile_all.sh {tag}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --no-cpu-baseline --no-bh --steps 10 --warmup 2",
})
for tree in ("host", "device"):
    fz, wz = counter(f"pmc_fetch_bh_{tree}", "k_bh_walk"), counter(f"pmc_write_bh_{tree}", "k_bh_walk")
    l1 = counter(f"pmc_l1_bh_{tree}", "k_bh_walk")
    b = bench(f"pmc_l1_bh_{tree}")
    visits = b["bh"]["node_visits_per_step"]
    write(f"pmc_traffic_bh_{tree}.json", {
        "kernel": last_kernel_name, "workload": f"configs[2] N=65536 Barnes-Hut theta=0.5, fast math, {tree} tree, 1 GPU", "commit": commit,
        "FETCH_SIZE_KiB_per_launch": fz, "WRITE_SIZE_KiB_per_launch": wz,
        "correction": "FETCH_SIZE doubled as for streaming reads; the walk's reads are divergent 16-B gathers, a width the guide calls uncalibrated, so treat the figure as +-2x",
        "hbm_bytes_per_launch": int(2 * fz * 1024 + wz * 1024),
        "TCP_TOTAL_CACHE_ACCESSES_per_launch": l1, "node_visits_per_launch": visits,
        "l1_line_accesses_per_visit": l1 / visits if visits else None,
        "note": "the node records are re-read ~1.2e8 times per launch from L1/L2; the writes are the per-segment partial planes",
        "source": f"tools/profile_all.sh {tag}: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, TCP_TOTAL_CACHE_ACCESSES_sum) --kernel-trace -- python3 bench.py --workload bh --tree {tree} --no-cpu-baseline --steps 10 --warmup 2; visits from that pass's own bench line",
    })
