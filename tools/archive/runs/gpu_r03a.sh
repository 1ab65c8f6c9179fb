#!/bin/bash
# guarded GEMM loads through range-checked buffer loads; dividing weight-gradient tiles
OUT=gpurun_out/r03a
mkdir -p $OUT
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_nn.py tests/test_gpu_model.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/tests.txt
python tools/bench_configs.py --steps 15 2>&1 | grep -v amdgpu | tee $OUT/configs.txt
python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c166-200
