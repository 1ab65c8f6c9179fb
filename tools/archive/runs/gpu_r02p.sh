#!/bin/bash
# r02p: double-staged GEMM (two K tiles in flight) -- parity, lab sweep, bench
mkdir -p gpurun_out/r02p
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_nn.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r02p/tests.txt
python tools/gemm_lab.py > gpurun_out/r02p/gemm_lab.txt 2>&1
tail -45 gpurun_out/r02p/gemm_lab.txt
python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 | tee gpurun_out/r02p/bench.json
