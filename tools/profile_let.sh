#!/bin/bash
# rocprofv3 kernel trace of the spatial-shard emulation (tools/let_report.py): per-kernel times of the five phases.
#   gpurun --timeout 600 -- 'bash tools/profile_let.sh r03 4194304'
set -eo pipefail
TAG=${1:-r03}
N=${2:-4194304}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
name=let_$N
rocprofv3 --kernel-trace --stats -d "$OUT/$name" -o "$name" -- python3 "$R/tools/let_report.py" --n "$N" --json "$OUT/$name.json" \
    > "$OUT/$name.txt" 2> "$OUT/$name.log"
python3 "$R/tools/rocpd_stats.py" "$(find "$OUT/$name" -name '*_results.db' | head -1)" > "$OUT/${name}_kernel_trace_stats.txt"
cat "$OUT/$name.txt"
head -40 "$OUT/${name}_kernel_trace_stats.txt"
