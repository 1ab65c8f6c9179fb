#!/bin/bash
# round-5 evidence beside the closing run: all five configurations, the C4 launch timeline, the edge-MLP micro-benchmark
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_r05_configs.sh <tag>'
TAG=${1:-r05m}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; ulimit -c 0
timeout 600 python tools/bench_configs.py --steps 20 > $OUT/configs.txt 2>&1; tail -6 $OUT/configs.txt
timeout 200 python tools/edge2_bench.py 2>&1 | grep "points, k" > $OUT/edge2_bench.txt; cat $OUT/edge2_bench.txt
bash tools/gpu_cfg_timeline.sh $TAG/c4 C4 > /dev/null 2>&1; tail -1 $OUT/c4/timeline.txt
