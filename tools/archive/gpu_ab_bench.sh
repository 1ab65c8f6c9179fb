#!/bin/bash
# un-profiled A/B of library builds on the bench step: tools/gpu_ab_bench.sh <tag> <reps> lib1.so lib2.so ...   (paths relative to the repo root)
TAG=$1; REPS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in $(seq $REPS); do for lib in "$@"; do
  name=$(basename $lib .so)
  DELTACONV_HIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/${name}_$rep.log 2>&1
  tail -1 $OUT/${name}_$rep.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', $rep, round(d['ms_per_step'],4), {k:v['us'] for k,v in d['roofline']['family'].items()})"
done; done | tee $OUT/summary.txt
