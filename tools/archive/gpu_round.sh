#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel trace.  Everything lands in
# gpurun_out/ (merged back by gpurun).  Usage: gpurun --timeout 1500 -- 'bash tools/gpu_round.sh [tag]'
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > $OUT/env.log 2>&1
rocm-smi --showclocks >> $OUT/env.log 2>&1
nproc >> $OUT/env.log; lscpu | grep -E "Model name|Socket|Core|Thread" >> $OUT/env.log
echo "== pytest -m gpu" 
timeout 900 python -m pytest tests -q -m gpu -rA --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; tail -2 $OUT/bench.log
echo "== kernels"
timeout 600 python tools/bench_kernels.py --json $OUT/kernels.json > $OUT/kernels.log 2>&1; tail -75 $OUT/kernels.log
echo "== rocprof"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_stats*" | head -3
python tools/prof_summary.py $OUT/prof > $OUT/kernel_summary.txt 2>&1; head -40 $OUT/kernel_summary.txt
python tools/step_timeline.py $OUT/prof > $OUT/step_timeline.txt 2>&1; tail -1 $OUT/step_timeline.txt
# keep the merged-back payload small
find $OUT/prof -name "*.csv" -size +20M -delete
