#!/bin/bash
OUT=gpurun_out/r02v
mkdir -p $OUT
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_nn.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/tests.txt
python tools/gemm_lab.py 2>&1 | grep "^tn\|^fwd embed\|^dX embed\|^fwd L3 hs" | cut -c1-150 | tee $OUT/lab_B.txt
for i in 1 2; do
DC_AB_LIB=tools/ab/libdeltaconv_hip_A.so python tools/ab_run.py bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c150-260
python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c150-260
done | tee $OUT/bench_ab.txt
python tools/bench_configs.py --steps 15 2>&1 | grep -v amdgpu | tee $OUT/configs.txt
