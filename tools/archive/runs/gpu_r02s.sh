#!/bin/bash
# r02s: weight-gradient GEMM through the LDS-staged kernel with the swept tile / slab plan
OUT=gpurun_out/r02s
mkdir -p $OUT
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_nn.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/tests.txt
python tools/gemm_lab.py 2>&1 | grep "^tn\|weight" | tee $OUT/gemm_lab_tn.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330 | tee $OUT/bench.json
python tools/bench_configs.py --steps 15 2>&1 | grep -v amdgpu | tee $OUT/configs.txt
