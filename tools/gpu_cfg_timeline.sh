#!/bin/bash
# ordered per-launch timeline + per-kernel summary of one configuration's eager train step
# usage: tools/gpu_cfg_timeline.sh <tag> <config substring>
TAG=$1; CFG=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp; ulimit -c 0
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --only "$CFG" --eager-only --steps 6 > $OUT/run.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof > $OUT/summary.txt 2>&1
python $GRAFT_REPO_ROOT/tools/step_timeline.py $OUT/prof > $OUT/timeline.txt 2>&1
find $OUT/prof -name "*.csv" -size +20M -delete
tail -2 $OUT/timeline.txt
