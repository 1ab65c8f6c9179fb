#!/bin/bash
# split-product GEMM: per-shape A/B, the GEMM-facing GPU tests, bench step A/B.   tools/gpu_x3.sh <tag>
TAG=${1:-x3}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 300 python tools/x3_ab.py > $OUT/x3_ab.txt 2>&1; tail -40 $OUT/x3_ab.txt
timeout 600 python -m pytest tests/test_gpu_nn.py tests/test_gpu_model.py -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
for e in 1 0 1 0; do
  DC_GEMM_EXACT=$e timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_exact$e.log 2>&1
  tail -1 $OUT/bench_exact$e.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('exact=$e', round(d['ms_per_step'],4), round(d['value'],1))"
done
