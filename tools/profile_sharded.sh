#!/bin/bash
# rocprofv3 kernel trace of one emulated rank of a world of G (tools/tune_sharded.py): the per-rank kernel chain of a
# sharded brute-force step.   gpurun --timeout 600 -- 'bash tools/profile_sharded.sh r02 8'
set -eo pipefail
TAG=${1:-r03}
G=${2:-8}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
name=sharded_g$G
rocprofv3 --kernel-trace --stats -d "$OUT/$name" -o "$name" -- python3 "$R/tools/tune_sharded.py" 65536 --g "$G" --steps 200 \
    > "$OUT/$name.txt" 2> "$OUT/$name.log"
python3 "$R/tools/rocpd_stats.py" "$(find "$OUT/$name" -name '*_results.db' | head -1)" > "$OUT/${name}_kernel_trace_stats.txt"
cat "$OUT/$name.txt"
head -20 "$OUT/${name}_kernel_trace_stats.txt"
