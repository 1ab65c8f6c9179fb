#!/bin/bash
OUT=gpurun_out/r02d
mkdir -p $OUT
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_lab.hip -o /tmp/mfma_lab && timeout 120 /tmp/mfma_lab > $OUT/mfma_lab.txt 2>&1; cat $OUT/mfma_lab.txt
timeout 300 python tools/knnmax_ab.py 2>&1 | grep -v amdgpu | tee $OUT/knnmax_ab.txt
timeout 300 python -m pytest tests/test_gpu_nn.py tests/test_gpu_geometry.py -q -x --tb=short -p no:cacheprovider 2>&1 | tail -4
