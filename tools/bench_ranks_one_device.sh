#!/bin/bash
# The multi-GPU bench rehearsed on ONE device: 4 rank processes over the one-device transport (control flow, not scaling).
# usage: tools/bench_ranks_one_device.sh tag count [bench args...]   (4 ranks, one device, ipc)
tag=$1; cnt=$2; shift 2
for i in $(seq 1 $cnt); do
  NBODY_BENCH_DEVICE=0 NBODY_TRANSPORT=ipc timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 4 --steps 5 --warmup 2 "$@" > gpurun_out/b4_${tag}_$i.json 2> gpurun_out/b4_${tag}_$i.err
  echo "rc=$?"
  python - <<PY
import json
try:
    b=json.load(open("gpurun_out/b4_${tag}_$i.json")); s=b.get("bh_spatial",{}); print("$tag", s.get("ms_per_step"), s.get("error"), [[round(x,1) for x in r["phase_ms"]] for r in s.get("per_rank",[])], (b.get("bf_large") or {}).get("ms_per_step"))
except Exception as e: print("$tag", "no json", e)
PY
done
