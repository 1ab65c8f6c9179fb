#!/bin/bash
# Round 4, first GPU call: transposed tile plan -- structure + bit identity, then the bench line (family_T) and the step timeline.
TAG=${1:-r04a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_tileT.py tests/test_gpu_tile.py tests/test_gpu_geometry.py tests/test_gpu_model.py -q --tb=short -p no:cacheprovider > $OUT/pytest_tileT.log 2>&1
tail -15 $OUT/pytest_tileT.log
grep -q passed $OUT/pytest_tileT.log && ! grep -q 'failed\|Aborted\|error' $OUT/pytest_tileT.log || exit 1
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log > $OUT/bench.json
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"])
r = d["roofline"]
for fam in ("family", "family_gather_path", "family_T", "family_T_gather_path"):
    print(fam, {k: (v["us"], v["frac"]) for k, v in (r.get(fam) or {}).items()})
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT/prof > $OUT/kernel_summary.txt 2>&1
python tools/step_timeline.py $OUT/prof > $OUT/step_timeline.txt 2>&1; tail -1 $OUT/step_timeline.txt
grep -E "tileT|ell_T|knn_max_bwd|edge_bwd|csc_|permute" $OUT/step_timeline.txt | cut -c1-150
find $OUT/prof -name "*.csv" -size +20M -delete
