#!/bin/bash
# Full GPU suite + smoke + bench line (with the CPU-baseline leg) + rocprofv3 kernel summary and ordered step timeline.
# usage: gpurun --timeout 2400 -- 'bash tools/gpu_full.sh <tag>'
TAG=${1:-r04}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -rA 2>&1 | grep -v "^PASSED" > $OUT/pytest_gpu.log
grep -E "passed|failed|pinned slots|scale 2\^" $OUT/pytest_gpu.log | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 500 python bench.py --steps 20 --warmup 5 ${BENCH_FLAGS:-} > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log > $OUT/bench.json
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "exact", d.get("exact_chain_ms_per_step"))
r = d["roofline"]
print("in_step", r.get("in_step"))
print("graded: frac", r.get("frac"), "us", r.get("us_per_launch"), "execution_only", r.get("execution_only"), "dispatch", r.get("dispatch"))
for name, rows in (r.get("in_step_kernels") or {}).items():
    print("  in-step", name, [(x["C"], x["us"], x["frac"]) for x in rows])
for fam in ("family", "family_gather_path", "family_T", "family_T_gather_path"):
    print(fam, {k: (v["us"], v["frac"]) for k, v in (r.get(fam) or {}).items()})
print("cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-chain --no-in-step-stamps > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT/prof > $OUT/kernel_summary.txt 2>&1
python tools/step_timeline.py $OUT/prof > $OUT/step_timeline.txt 2>&1; tail -1 $OUT/step_timeline.txt
find $OUT/prof -name "*.csv" -size +20M -delete
