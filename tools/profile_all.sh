#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/: a kernel trace of the default bench line and of the
# two Barnes-Hut variants, then FETCH_SIZE / WRITE_SIZE / TCP_TOTAL_CACHE_ACCESSES in separate --pmc passes (never
# together with a trace other than --kernel-trace).  Run on the GPU box from the repo root:
#   gpurun --timeout 1100 -- 'bash tools/profile_all.sh r02'
# Outputs land in gpurun_out/prof_<tag>/ as text; tools/make_pmc_json.py turns them into profiles/*.json.
set -eo pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp

trace() {  # name, bench args...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats -d "$OUT/$name" -o "$name" -- python3 "$R/bench.py" --no-cpu-baseline --no-bh "$@" \
        > "$OUT/$name.bench.json" 2> "$OUT/$name.log"
    python3 "$R/tools/rocpd_stats.py" "$(find "$OUT/$name" -name '*_results.db' | head -1)" > "$OUT/${name}_kernel_trace_stats.txt"
    echo "== $name"; head -8 "$OUT/${name}_kernel_trace_stats.txt"
}
pmc() {  # name, counter, bench args...
    local name=$1 counter=$2; shift 2
    rocprofv3 --pmc "$counter" --kernel-trace -d "$OUT/$name" -o "$name" -- python3 "$R/bench.py" --no-cpu-baseline --no-bh --steps 10 --warmup 2 "$@" \
        > "$OUT/$name.bench.json" 2> "$OUT/$name.log"
    { echo "== rocprofv3 --pmc $counter: $name"; python3 "$R/tools/rocpd_pmc.py" "$(find "$OUT/$name" -name '*_results.db' | head -1)" "${counter%% *}"; } >> "$OUT/pmc_summary.txt"
}

trace bf65536
trace bh65536_host --workload bh --tree host
trace bh65536_device --workload bh --tree device
python3 "$R/tools/rocpd_timeline.py" "$(find "$OUT/bh65536_device" -name '*_results.db' | head -1)" k_tree_keys 120 > "$OUT/bh65536_device_step_timeline.txt" || true
: > "$OUT/pmc_summary.txt"
pmc pmc_fetch_bf FETCH_SIZE
pmc pmc_write_bf WRITE_SIZE
pmc pmc_fetch_bh_host FETCH_SIZE --workload bh --tree host
pmc pmc_write_bh_host WRITE_SIZE --workload bh --tree host
pmc pmc_l1_bh_host TCP_TOTAL_CACHE_ACCESSES_sum --workload bh --tree host
pmc pmc_fetch_bh_device FETCH_SIZE --workload bh --tree device
pmc pmc_write_bh_device WRITE_SIZE --workload bh --tree device
pmc pmc_l1_bh_device TCP_TOTAL_CACHE_ACCESSES_sum --workload bh --tree device
cat "$OUT/pmc_summary.txt"
