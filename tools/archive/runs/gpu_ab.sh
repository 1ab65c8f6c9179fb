#!/bin/bash
# A/B of two builds of the library on the GEMM lab: A = tools/ab/libdeltaconv_hip_A.so, B = in-tree; A runs twice
OUT=gpurun_out/${1:-ab}
mkdir -p $OUT
DC_AB_LIB=tools/ab/libdeltaconv_hip_A.so python tools/gemm_lab.py 2>&1 | grep -v amdgpu | cut -c1-150 > $OUT/lab_A1.txt
python tools/gemm_lab.py 2>&1 | grep -v amdgpu | cut -c1-150 > $OUT/lab_B.txt
DC_AB_LIB=tools/ab/libdeltaconv_hip_A.so python tools/gemm_lab.py 2>&1 | grep -v amdgpu | cut -c1-150 > $OUT/lab_A2.txt
paste -d'|' <(cut -c1-34,45-54 $OUT/lab_A1.txt) <(cut -c45-54 $OUT/lab_B.txt) <(cut -c45-54 $OUT/lab_A2.txt) <(cut -c35-44 $OUT/lab_B.txt)
