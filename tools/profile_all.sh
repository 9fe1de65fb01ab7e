#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/: a kernel trace of the default bench line and of the
# two Barnes-Hut variants, then FETCH_SIZE / WRITE_SIZE in separate --pmc passes (never together with a
# trace other than --kernel-trace). Run on the GPU box from the repo root:
#   gpurun --timeout 900 -- 'bash tools/profile_all.sh r01'
# Outputs land in gpurun_out/prof_<tag>/ as text; copy the ones to keep into profiles/.
set -eo pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp

trace() {  # name, bench args...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats -d "$OUT/$name" -o "$name" -- python3 "$R/bench.py" --no-cpu-baseline "$@" \
        > "$OUT/$name.bench.json" 2> "$OUT/$name.log"
    python3 "$R/tools/rocpd_stats.py" "$(find "$OUT/$name" -name '*_results.db' | head -1)" > "$OUT/${name}_kernel_trace_stats.txt"
    echo "== $name"; head -8 "$OUT/${name}_kernel_trace_stats.txt"
}
pmc() {  # name, counter, bench args...
    local name=$1 counter=$2; shift 2
    rocprofv3 --pmc "$counter" --kernel-trace -d "$OUT/$name" -o "$name" -- python3 "$R/bench.py" --no-cpu-baseline --steps 10 --warmup 2 "$@" \
        > /dev/null 2> "$OUT/$name.log"
    { echo "== rocprofv3 --pmc $counter: $name"; python3 "$R/tools/rocpd_pmc.py" "$(find "$OUT/$name" -name '*_results.db' | head -1)" "$counter"; } >> "$OUT/pmc_fetch_write_size.txt"
}

trace bf65536
trace bh65536_host --workload bh --tree host
trace bh65536_device --workload bh --tree device
: > "$OUT/pmc_fetch_write_size.txt"
pmc pmc_fetch_bf FETCH_SIZE
pmc pmc_write_bf WRITE_SIZE
pmc pmc_fetch_bh FETCH_SIZE --workload bh --tree host
pmc pmc_write_bh WRITE_SIZE --workload bh --tree host
pmc pmc_fetch_bhdev FETCH_SIZE --workload bh --tree device
pmc pmc_write_bhdev WRITE_SIZE --workload bh --tree device
cat "$OUT/pmc_fetch_write_size.txt"
