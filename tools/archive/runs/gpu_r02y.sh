#!/bin/bash
# r02y: grad_T with folded accumulation, wave-per-column CSC rank kernel
OUT=gpurun_out/r02y
mkdir -p $OUT
python -m pytest tests/test_gpu_geometry.py tests/test_gpu_nn.py tests/test_gpu_model.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/tests.txt
for i in 1 2; do
DC_AB_LIB=tools/ab/libdeltaconv_hip_A.so python tools/ab_run.py bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c166-200
python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c166-200
done | tee $OUT/bench_ab.txt
python tools/bench_configs.py --steps 15 2>&1 | grep -v amdgpu | tee $OUT/configs.txt
