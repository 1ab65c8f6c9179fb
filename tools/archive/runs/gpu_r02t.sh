#!/bin/bash
# r02t: GEMM main loop with two K tiles in flight, LDS-only barrier, exact sched_group_barrier plan
OUT=gpurun_out/r02t
mkdir -p $OUT
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_nn.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/tests.txt
python tools/gemm_lab.py > $OUT/gemm_lab.txt 2>&1
grep -v amdgpu $OUT/gemm_lab.txt | cut -c1-150
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330 | tee $OUT/bench.json
