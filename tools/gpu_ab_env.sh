#!/bin/bash
# Same-box A/B of an environment switch on the bench line: usage  gpurun -- 'bash tools/gpu_ab_env.sh <tag> VAR=a VAR=b [bench flags]'
TAG=$1; A=$2; B=$3; shift 3
OUT=gpurun_out/$TAG; mkdir -p $OUT
for round in 1 2; do
for spec in "$A" "$B"; do
  env $spec python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-exact-chain "$@" > $OUT/bench_${spec//[^A-Za-z0-9]/_}_$round.log 2>&1
  tail -1 $OUT/bench_${spec//[^A-Za-z0-9]/_}_$round.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$spec round $round: ms', round(d['ms_per_step'],4), '| frac', r['frac'], 'us', r['us_per_launch'], 'rot', (r.get('rotating_buffers') or {}).get('us_per_launch'), 'l3', r.get('us_per_launch_l3_resident'))
for fam in ('family','family_T'):
    print('   ', fam, {k:v['us'] for k,v in r[fam].items()})
for name, rows in (r.get('in_step_kernels') or {}).items():
    print('    in-step', name, [(x['C'], x['us']) for x in rows])
"
done
done
