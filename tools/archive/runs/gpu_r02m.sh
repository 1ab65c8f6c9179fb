#!/bin/bash
OUT=gpurun_out/r02m
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_nn.py tests/test_gpu_dist.py -q -m gpu --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
tail -4 $OUT/pytest_gpu.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT/prof > $OUT/kernel_summary.txt 2>&1
python tools/step_timeline.py $OUT/prof > $OUT/step_timeline.txt 2>&1; tail -1 $OUT/step_timeline.txt
find $OUT/prof -name "*.csv" -size +20M -delete
