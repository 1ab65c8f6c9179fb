#!/bin/bash
# PMC passes over the ELL apply kernels (each pass in its own rocprofv3 run: --pmc + --kernel-trace only).
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -ciE "FETCH_SIZE|TCC_HIT|SQ_WAIT_INST_ANY|TCP_" $OUT/counters_list.txt
run() {  # name, counters..., then -- driver args
  name=$1; shift; ctrs=(); while [ "$1" != "--" ]; do ctrs+=($1); shift; done; shift
  timeout 300 rocprofv3 --kernel-trace --pmc ${ctrs[@]} --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/tools/apply_driver.py "$@" > $OUT/$name.log 2>&1
  echo "$name rc=$?"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/$name | tee $OUT/$name.summary.txt
  find $OUT/$name -name "*.csv" -size +5M -delete
}
# PMC_CFGS="c2" limits the passes to the graded size through the tiled kernels (the `traffic` figure of the bench line)
for cfg in "c2 --batch 32" "c2gather --batch 32 --gather" "big --batch 512 --iters 5" "biggather --batch 512 --iters 5 --gather"; do
  set -- $cfg; tag=$1; shift
  if [ -n "$PMC_CFGS" ] && [[ " $PMC_CFGS " != *" $tag "* ]]; then continue; fi
  run ${tag}_fetch FETCH_SIZE -- "$@"
  run ${tag}_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- "$@"
  run ${tag}_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM -- "$@"
  run ${tag}_tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr -- "$@"
  run ${tag}_lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- "$@"
done
