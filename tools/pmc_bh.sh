#!/bin/bash
# L1/L2 counters of the Barnes-Hut walk (one --pmc pass per counter group, kernel trace only).
#   gpurun --timeout 600 -- 'bash tools/pmc_bh.sh'
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_bh${NBODY_BH_VARIANT:-0}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
: > "$OUT/summary.txt"
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" "SQ_WAVES SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
    name=$(echo $grp | tr ' ' '_' | cut -c1-40)
    if rocprofv3 --pmc $grp --kernel-trace -d "$OUT/$name" -o x -- python3 "$R/bench.py" --no-cpu-baseline --workload bh --tree device --steps 5 --warmup 2 > /dev/null 2> "$OUT/$name.log"; then
        { echo "== $grp"; python3 "$R/tools/rocpd_pmc.py" "$(find "$OUT/$name" -name '*_results.db' | head -1)" | grep -i "k_bh_walk\|kernel " ; } >> "$OUT/summary.txt"
    else
        echo "== $grp: FAILED ($(tail -1 "$OUT/$name.log"))" >> "$OUT/summary.txt"
    fi
done
cat "$OUT/summary.txt"
