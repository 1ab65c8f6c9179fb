#!/bin/bash
OUT=gpurun_out/r02e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_gemm.py -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
timeout 600 python tools/gemm_lab.py > $OUT/gemm_lab.txt 2>&1; cat $OUT/gemm_lab.txt | grep -v amdgpu.ids
timeout 300 python tools/gemm_ablate.py 2>&1 | grep -v amdgpu | tee $OUT/gemm_ablate.txt
