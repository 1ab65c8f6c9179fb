#!/bin/bash
# A/B: tools/ab/libdeltaconv_hip_A.so vs in-tree library -- GEMM lab (A, B, A) and the bench (A, B, A, B)
OUT=gpurun_out/${1:-ab}
mkdir -p $OUT
python -m pytest tests/test_gpu_gemm.py -x -q -m gpu 2>&1 | tail -2
DC_AB_LIB=tools/ab/libdeltaconv_hip_A.so python tools/gemm_lab.py 2>&1 | grep -v amdgpu | cut -c1-150 > $OUT/lab_A1.txt
python tools/gemm_lab.py 2>&1 | grep -v amdgpu | cut -c1-150 > $OUT/lab_B.txt
DC_AB_LIB=tools/ab/libdeltaconv_hip_A.so python tools/gemm_lab.py 2>&1 | grep -v amdgpu | cut -c1-150 > $OUT/lab_A2.txt
paste -d'|' <(grep "^fwd\|^dX" $OUT/lab_A1.txt | cut -c1-34,45-54) <(grep "^fwd\|^dX" $OUT/lab_B.txt | cut -c45-54) <(grep "^fwd\|^dX" $OUT/lab_A2.txt | cut -c45-54) <(grep "^fwd\|^dX" $OUT/lab_B.txt | cut -c35-44)
paste -d'|' <(grep "^tn" $OUT/lab_A1.txt | cut -c1-34,55-64) <(grep "^tn" $OUT/lab_B.txt | cut -c55-64) <(grep "^tn" $OUT/lab_A2.txt | cut -c55-64)
for i in 1 2; do
DC_AB_LIB=tools/ab/libdeltaconv_hip_A.so python tools/ab_run.py bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c166-200
python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c166-200
done | tee $OUT/bench_ab.txt
