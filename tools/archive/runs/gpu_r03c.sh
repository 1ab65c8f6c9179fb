#!/bin/bash
OUT=gpurun_out/r03c
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_nn.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/tests.txt
python tools/bench_configs.py --steps 15 --only C4 2>&1 | grep -v amdgpu | tee $OUT/configs.txt
python tools/bench_configs.py --steps 15 --only C5 2>&1 | grep -v amdgpu | tail -1 | tee -a $OUT/configs.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_C4 -o p -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --only C4 --eager-only --steps 8 > $GRAFT_REPO_ROOT/$OUT/rocprof_C4.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT/prof_C4 > $OUT/C4_kernel_summary.txt 2>&1
python tools/step_timeline.py $OUT/prof_C4 > $OUT/C4_step_timeline.txt 2>&1; tail -1 $OUT/C4_step_timeline.txt
find $OUT -name "*.csv" -size +20M -delete
grep -n "gemm_kernel" $OUT/C4_step_timeline.txt | sed -e 's/(anonymous namespace):://g; s/void //' | cut -c1-100 | awk '$1>50'
