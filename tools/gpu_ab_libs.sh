#!/bin/bash
# Same-box A/B of several builds of the library (tools/ab/*.so) on the in-step gap lab and on the bench step, two rounds.
# usage: gpurun -- 'bash tools/gpu_ab_libs.sh <tag> lib1.so lib2.so ...'   ("default" = the in-tree library)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for round in 1 2; do
for lib in "$@"; do
  if [ "$lib" = default ]; then unset DELTACONV_HIP_LIB; else export DELTACONV_HIP_LIB=$PWD/$lib; fi
  echo "=== round $round lib: $lib"
  [ $round = 1 ] && python tools/instep_gap_lab.py 2>&1 | grep -E "^A|^B|^C"
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-exact-chain 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('  bench ms', round(d['ms_per_step'],4), 'graded exec', (r.get('execution_only') or {}).get('us'))
print('   ', ' '.join(f'{n}:' + '/'.join(str(x['us']) for x in rows) for n, rows in (r.get('in_step_kernels') or {}).items()))
"
done
done 2>&1 | tee $OUT/ab_libs.txt
