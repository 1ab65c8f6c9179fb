#!/bin/bash
OUT=gpurun_out/exp_sorted; mkdir -p $OUT
python tools/bench_kernels.py --channels 64 > $OUT/unsorted.log 2>&1
python tools/bench_kernels.py --channels 64 --sorted > $OUT/sorted.log 2>&1
paste <(grep -E "^(apply|knn_max|knn )" $OUT/unsorted.log | grep "remap': 1" | awk '{print $1, $2}') <(grep -E "^(apply|knn_max|knn )" $OUT/sorted.log | grep "remap': 1" | awk '{print $2}')
grep -E "^knn|^mls|^csc" $OUT/sorted.log
