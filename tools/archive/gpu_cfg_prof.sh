#!/bin/bash
# per-kernel summary of one configuration's eager train step with and without the tile plan
# usage: tools/gpu_cfg_prof.sh <tag> <config substring>
TAG=$1; CFG=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for v in 1 0; do
  DC_TILE_PLAN=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -o t -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --only "$CFG" --eager-only --steps 10 > $OUT/run_$v.log 2>&1
  python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof_$v > $OUT/summary_tile$v.txt 2>&1
  find $OUT/prof_$v -name "*.csv" -size +20M -delete
done
