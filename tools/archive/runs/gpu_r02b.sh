#!/bin/bash
OUT=gpurun_out/r02b
mkdir -p $OUT
export TMPDIR=/tmp
echo "== gemm lab"
timeout 600 python tools/gemm_lab.py > $OUT/gemm_lab.txt 2>&1; cat $OUT/gemm_lab.txt | grep -v amdgpu.ids
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
tail -25 $OUT/pytest_gpu.log
