#!/bin/bash
# r02z: kernel summaries of the C3 / C4 / C5 eager train steps on the round-2 final kernels
OUT=gpurun_out/r02z
mkdir -p $OUT
export TMPDIR=/tmp
for C in C3 C4 C5; do
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$C -o p -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --only $C --eager-only --steps 8 > $GRAFT_REPO_ROOT/$OUT/rocprof_$C.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT/prof_$C > $OUT/${C}_kernel_summary.txt 2>&1
python tools/step_timeline.py $OUT/prof_$C > $OUT/${C}_step_timeline.txt 2>&1; tail -1 $OUT/${C}_step_timeline.txt
done
find $OUT -name "*.csv" -size +20M -delete
head -45 $OUT/C5_kernel_summary.txt | cut -c1-125
