#!/bin/bash
# full GPU suite on the split-product default + bench A/B against the exact chain.   tools/gpu_x3_full.sh <tag>
TAG=${1:-x3full}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; tail -15 $OUT/pytest.txt
for e in 1 0 1 0; do
  DC_GEMM_EXACT=$e timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_exact$e.log 2>&1
  tail -1 $OUT/bench_exact$e.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('exact=$e', round(d['ms_per_step'],4), round(d['value'],1))"
done
