#!/bin/bash
# Data-parallel proxy on ONE GPU (no multi-GPU node is available to the build): bench.py through the launcher with
# --force-dist (RCCL process group, the collective path of the step: replay -> all-reduce -> replay) against the
# single-process step, at the per-rank batches of 8-GPU runs.  usage: tools/gpu_dp_proxy.sh <tag>
TAG=${1:-dp}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # label batch points k
  local label=$1 b=$2 n=$3 k=$4
  local a=$(python bench.py --steps 40 --warmup 5 --no-cpu-baseline --batch $b --points $n --k $k 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
  local d=$(python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --force-dist --batch $b --points $n --k $k 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
  local e=$(python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --force-dist --no-graph --batch $b --points $n --k $k 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
  echo "$label | clouds/rank $b x $n pts k=$k | single process (1 graph) $a ms | --force-dist (graph, all-reduce, graph) $d ms | --force-dist eager $e ms"
}
{
echo "# ModelNet40 classification net (bench.py), per-rank shapes of an 8-GPU data-parallel run; ms per step on one MI355X"
run "C2 per rank (weak scaling: 32 clouds/GPU)" 32 1024 20
run "C2 / 8 (strong scaling of the 32-cloud batch)" 4 1024 20
run "C4-like per rank (2 clouds of 2048 points)" 2 2048 20
run "C5-like per rank (k = 30, 4096 points; 2 clouds: a 1-row BatchNorm cannot train)" 2 4096 30
} | tee $OUT/dp_proxy.txt
