#!/bin/bash
# r03d: round-2 closing run -- full GPU suite, bench line, rocprof kernel summary of the bench command, step
# timeline, GEMM counters (own vs library), all configurations
OUT=gpurun_out/r03d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
tail -4 $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log > $OUT/bench.json; cut -c1-200 $OUT/bench.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT/prof > $OUT/kernel_summary.txt 2>&1
python tools/step_timeline.py $OUT/prof > $OUT/step_timeline.txt 2>&1; tail -1 $OUT/step_timeline.txt
find $OUT/prof -name "*.csv" -size +20M -delete
PMC_TAG=r03d/pmc_gemm bash tools/gpu_pmc_gemm2.sh 2>&1 | tee $OUT/pmc_gemm_summary.txt
python tools/bench_configs.py --steps 15 2>&1 | grep -v amdgpu | tee $OUT/configs.txt
