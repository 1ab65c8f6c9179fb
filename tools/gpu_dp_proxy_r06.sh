#!/bin/bash
# Per-rank proxies of the strong-scaling configurations on ONE GPU (round 6): bench.py --config/--global-batch = the per-rank share
# of an 8-GPU run, graph-replayed step, + a rocprofv3 kernel trace of the same command for the launch count and the ordered timeline.
# usage: gpurun -- 'bash tools/gpu_dp_proxy_r06.sh <tag>'
TAG=${1:-r06dp}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; ulimit -c 0
cd $GRAFT_REPO_ROOT
{
echo "# per-rank steps of 8-GPU strong scaling on one MI355X: ms per graph-replayed train step (per-rank BatchNorm | synchronised BatchNorm, RCCL group of one rank), launches per step"
for spec in "C2 4 C2/8_(4x1024,k=20)" "C4 2 C4/8_(2x2048,k=20)" "C5 1 C5/8_(1x4096,k=30)" "C2 32 C2_full" "C4 16 C4_full" "C5 8 C5_full" "C3 32 C3_full"; do
  set -- $spec
  a=$(python bench.py --config $1 --global-batch $2 --steps 40 --warmup 5 --no-cpu-baseline --no-exact-chain --no-in-step-stamps 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
  b=""
  if [ "$2" -le 4 ] && [ "$1" != "C5" -o "$2" -ge 2 ]; then
    b=$(python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29581 bench.py --gpus 1 --config $1 --global-batch $2 --steps 40 --warmup 5 --no-cpu-baseline --no-exact-chain --force-dist 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
  fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_$1_$2 -o t -- python $GRAFT_REPO_ROOT/bench.py --config $1 --global-batch $2 --steps 6 --warmup 3 --no-cpu-baseline --no-exact-chain --no-in-step-stamps > $OUT/trace_$1_$2.log 2>&1)
  l=$(python tools/step_timeline.py $OUT/prof_$1_$2 2>/dev/null | tail -1)
  python tools/step_timeline.py $OUT/prof_$1_$2 > $OUT/timeline_$1_$2.txt 2>&1
  find $OUT/prof_$1_$2 -name "*.csv" -delete
  echo "$3: $a ms | sync-BN one graph: ${b:-n/a} ms | $l"
done
} | tee $OUT/dp_proxy.txt
