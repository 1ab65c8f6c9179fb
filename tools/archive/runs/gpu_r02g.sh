#!/bin/bash
OUT=gpurun_out/r02g
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/bench_configs.py --steps 15 --tuned 2>&1 | grep -v amdgpu | tee $OUT/configs.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_c4 -o c4 -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --only C4 --eager-only --steps 8 --tuned > $GRAFT_REPO_ROOT/$OUT/rocprof_c4.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT/prof_c4 > $OUT/c4_kernel_summary.txt 2>&1; head -50 $OUT/c4_kernel_summary.txt
find $OUT -name "*.csv" -size +20M -delete
