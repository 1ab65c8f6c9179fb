#!/bin/bash
# Step timelines (rocprofv3 kernel trace) of the bench step under two settings of an environment switch, same box.
# usage: gpurun -- 'bash tools/gpu_timeline_ab.sh <tag> VAR=a VAR=b [bench flags]'
TAG=$1; A=$2; B=$3; shift 3
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for spec in "$A" "$B"; do
  name=${spec//[^A-Za-z0-9]/_}
  (cd /tmp && env $spec timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$name -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-chain --no-in-step-stamps "$@" > $GRAFT_REPO_ROOT/$OUT/rocprof_$name.log 2>&1)
  python tools/step_timeline.py $OUT/prof_$name > $OUT/step_timeline_$name.txt 2>&1
  tail -1 $OUT/step_timeline_$name.txt
  rm -rf $OUT/prof_$name
done
