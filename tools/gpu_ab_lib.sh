#!/bin/bash
# A/B of two builds of the library on the bench step: rocprofv3 kernel traces -> ordered step timelines -> diff
# usage: tools/gpu_ab_lib.sh <tag> <other-lib.so (relative to the repo root)>
TAG=$1; OTHER=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_a -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/a.log 2>&1
DELTACONV_HIP_LIB=$GRAFT_REPO_ROOT/$OTHER timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_b -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/b.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/step_timeline.py $OUT/prof_a > $OUT/timeline_a.txt 2>&1; tail -1 $OUT/timeline_a.txt
python tools/step_timeline.py $OUT/prof_b > $OUT/timeline_b.txt 2>&1; tail -1 $OUT/timeline_b.txt
python tools/timeline_diff.py $OUT/timeline_a.txt $OUT/timeline_b.txt > $OUT/diff.txt 2>&1
find $OUT -name "*.csv" -size +20M -delete
