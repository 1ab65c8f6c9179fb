#!/bin/bash
# ordered step timeline + kernel summary of the bench command: tools/gpu_timeline.sh <tag>
TAG=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT/prof > $OUT/kernel_summary.txt 2>&1
python tools/step_timeline.py $OUT/prof > $OUT/step_timeline.txt 2>&1; tail -1 $OUT/step_timeline.txt
find $OUT/prof -name "*.csv" -size +20M -delete
