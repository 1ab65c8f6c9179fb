#!/bin/bash
# PMC passes over the BACKWARD family (transposed applies + max-aggregation backward), tiled (transposed plan) and gather path;
# each pass its own rocprofv3 run (--pmc + --kernel-trace only).  usage: tools/pmc_applyT.sh <tag>
TAG=${1:-pmcT}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
run() {  # name, counters..., then -- driver args
  name=$1; shift; ctrs=(); while [ "$1" != "--" ]; do ctrs+=($1); shift; done; shift
  timeout 300 rocprofv3 --kernel-trace --pmc ${ctrs[@]} --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/tools/apply_driver.py --transposed --iters 20 "$@" > $OUT/$name.log 2>&1
  echo "$name rc=$?"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/$name | tee $OUT/$name.summary.txt
  find $OUT/$name -name "*.csv" -size +5M -delete
}
run c2T_fetch FETCH_SIZE -- --batch 32
run c2T_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- --batch 32
run c2T_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM -- --batch 32
run c2T_tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr -- --batch 32
run c2T_lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- --batch 32
run c2Tgather_fetch FETCH_SIZE -- --batch 32 --gather
run c2Tgather_tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr -- --batch 32 --gather
