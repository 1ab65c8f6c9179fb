#!/bin/bash
# Round-3 closing run: full GPU suite, smoke, bench line, rocprofv3 kernel summary + ordered step timeline of the bench
# command, counters of the graded apply kernel (separate --pmc passes), all configurations.
# usage: gpurun --timeout 3000 -- 'bash tools/gpu_round3.sh <tag>'
TAG=${1:-r03z}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log > $OUT/bench.json; cut -c1-200 $OUT/bench.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT/prof > $OUT/kernel_summary.txt 2>&1
python tools/step_timeline.py $OUT/prof > $OUT/step_timeline.txt 2>&1; tail -1 $OUT/step_timeline.txt
find $OUT/prof -name "*.csv" -size +20M -delete
PMC_CFGS="${PMC_CFGS:-c2 c2gather}" bash tools/pmc_apply.sh $TAG/pmc_apply > $OUT/pmc_apply_summary.txt 2>&1; grep -c . $OUT/pmc_apply_summary.txt
python tools/bench_configs.py --steps 20 2>&1 | grep -v amdgpu | tee $OUT/configs.txt
bash tools/gpu_x3_pmc.sh $TAG/x3pmc > $OUT/x3pmc_summary.txt 2>&1; grep -c gemm_kernel $OUT/x3pmc_summary.txt
