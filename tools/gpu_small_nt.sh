#!/bin/bash
# Per-rank proxies of the strong-scaling configurations (C4: 2 x 2048 points, C5: 1 x 4096 points k = 30, C2 / 8: 4 x 1024) with the
# tile plans forced on (DC_TILE_P=64 / 32) against the policy default (gather path below 8192 points).  usage: gpurun -- 'bash tools/gpu_small_nt.sh <tag>'
TAG=${1:-r06nt}
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() {
  local label="$1"; shift
  local ms=$(env "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))")
  echo "$label: $ms ms"
}
for round in 1 2; do
for cfg in "C4 2" "C5 1" "C2 4"; do
  set -- $cfg
  for P in 0 64 32; do
    run "$1 global-batch $2 DC_TILE_P=$P (round $round)" DC_TILE_P=$P python bench.py --config $1 --global-batch $2 --steps 40 --warmup 5 --no-cpu-baseline --no-exact-chain
  done
done
done | tee $OUT/small_nt.txt
