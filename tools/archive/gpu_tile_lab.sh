#!/bin/bash
# tile lab on the GPU box: timings at B = 32 / 512, then counters (separate --pmc passes) at B = 32
TAG=${1:-r03a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 tools/tile_lab.bin 32 512 > $OUT/tile_lab.txt 2>&1; echo "lab rc=$?"
cat $OUT/tile_lab.txt
if [ "$2" == "pmc" ]; then
cd /tmp
pass() { name=$1; shift
  TILE_LAB_ITERS=8 timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- $GRAFT_REPO_ROOT/tools/tile_lab.bin ${PMC_B:-32} > $OUT/$name.log 2>&1
  echo "$name rc=$?"; python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $OUT/$name k_ > $OUT/$name.summary.txt
  find $OUT/$name -name "*.csv" -size +5M -delete
}
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
pass fetch FETCH_SIZE
pass lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
fi
