#!/bin/bash
OUT=gpurun_out/cfg; mkdir -p $OUT
for t in C2_modelnet40 C3_scanobjectnn C4_shapenet C5_shapeseg; do
  timeout 600 python -m pytest "tests/test_gpu_configs.py::test_full_size_properties[$t]" -q -m gpu -x --tb=short -p no:cacheprovider > $OUT/$t.log 2>&1
  echo "$t rc=$?"; grep -E "passed|failed|Error|error|fault|Fatal|File \"/root/repo" $OUT/$t.log | head -8
done
timeout 900 python -m pytest tests/test_gpu_configs.py -k reduced -q -m gpu --tb=short -p no:cacheprovider > $OUT/reduced.log 2>&1; tail -5 $OUT/reduced.log
