#!/bin/bash
# Round 6 quick check: the tests touched this round + the bench line with the in-step stamps + per-rank proxies of the
# strong-scaling configurations.  usage: gpurun --timeout 1500 -- 'bash tools/gpu_r06_quick.sh <tag> [pytest -k expression]'
TAG=${1:-r06q}
KEXPR=${2:-"stamps or pinned or mls_stages or stats_mode or full_size"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ulimit -c 0
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -k "$KEXPR" 2>&1 | tail -25 > $OUT/pytest_quick.log
tail -5 $OUT/pytest_quick.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log > $OUT/bench.json
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
r = d["roofline"]
print("C2 ms_per_step", round(d["ms_per_step"], 4), "exact", d.get("exact_chain_ms_per_step"))
print("frac", r["frac"], "us", r["us_per_launch"], "execution_only", r.get("execution_only"), "dispatch", r.get("dispatch"), "rotating", r.get("rotating_buffers"), "l3", r.get("us_per_launch_l3_resident"))
for name, rows in (r.get("in_step_kernels") or {}).items():
    print("  in-step", name, [(x["C"], x["us"], x["frac"]) for x in rows])
PY
for spec in "C4 2" "C4 16" "C5 1" "C5 8" "C3 32"; do
  set -- $spec
  timeout 300 python bench.py --config $1 --global-batch $2 --steps 20 --warmup 5 --no-cpu-baseline --no-exact-chain > $OUT/bench_$1_$2.log 2>&1
  tail -1 $OUT/bench_$1_$2.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('$1 global-batch $2:', round(d['ms_per_step'],4), 'ms', round(d['value'],1), 'clouds/s', d['scaling'], '| frac', r['frac'], 'us', r['us_per_launch'], '|', d['config']['workload'][:90])
except Exception as e:
    print('$1 $2 FAILED', e)
"
done
