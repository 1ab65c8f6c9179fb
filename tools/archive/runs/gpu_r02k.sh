#!/bin/bash
OUT=gpurun_out/r02k
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
tail -6 $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log
timeout 600 python tools/bench_configs.py --steps 15 2>&1 | grep -v amdgpu | tee $OUT/configs.txt
