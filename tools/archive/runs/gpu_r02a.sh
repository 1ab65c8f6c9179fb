#!/bin/bash
# r02a: parity tests (old + new), gather lab 2, GEMM lab, baseline bench
OUT=gpurun_out/r02a
mkdir -p $OUT
export TMPDIR=/tmp
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gather_lab2.hip -o /tmp/gl2 && timeout 400 /tmp/gl2 > $OUT/gather_lab2.txt 2>&1 ) &
LABPID=$!
wait $LABPID
tail -60 $OUT/gather_lab2.txt
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
tail -15 $OUT/pytest_gpu.log
echo "== gemm lab"
timeout 400 python tools/gemm_lab.py > $OUT/gemm_lab.txt 2>&1; cat $OUT/gemm_lab.txt
echo "== bench"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1; tail -2 $OUT/bench.log
