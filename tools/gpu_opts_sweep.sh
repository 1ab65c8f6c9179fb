#!/bin/bash
# Same-box sweep of DC_OPTIONS settings on one bench form: usage  gpurun -- 'bash tools/gpu_opts_sweep.sh <tag> "<bench flags>" opt1 opt2 ...'
# (an option string is a DC_OPTIONS value, e.g. "5=3,6=16"; "-" = none).  Two rounds, ms per step.
TAG=$1; FLAGS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for round in 1 2; do
for o in "$@"; do
  v=$o; [ "$o" = "-" ] && v=""
  ms=$(DC_OPTIONS="$v" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-exact-chain --no-in-step-stamps $FLAGS 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
  echo "round $round  DC_OPTIONS='$v'  $FLAGS : $ms ms"
done
done | tee $OUT/sweep.txt
