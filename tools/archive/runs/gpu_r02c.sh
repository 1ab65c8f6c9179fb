#!/bin/bash
OUT=gpurun_out/r02c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_nn.py -q -x --tb=short -p no:cacheprovider 2>&1 | tail -5
timeout 600 python tools/gemm_lab.py > $OUT/gemm_lab.txt 2>&1; cat $OUT/gemm_lab.txt | grep -v amdgpu.ids
timeout 600 python tools/gemm_lab.py 1 > $OUT/gemm_lab_phase.txt 2>&1; cat $OUT/gemm_lab_phase.txt | grep -v amdgpu.ids
