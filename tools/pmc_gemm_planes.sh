#!/bin/bash
# SQ counters of the forward / input-gradient products with the weight operand from pre-split planes vs split in the K loop
# (two --pmc passes each, --kernel-trace only beside them).   gpurun -- 'bash tools/pmc_gemm_planes.sh <tag>'
TAG=${1:-pmc_planes}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; ulimit -c 0
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $OUT/${name}_p1 -o p -- python $GRAFT_REPO_ROOT/tools/gemm_planes_driver.py > $OUT/${name}_p1.log 2>&1
  env "$@" timeout 200 rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $OUT/${name}_p2 -o p -- python $GRAFT_REPO_ROOT/tools/gemm_planes_driver.py > $OUT/${name}_p2.log 2>&1
}
run planes DC_NO_PLANES=0
run inloop DC_NO_PLANES=1
cd $GRAFT_REPO_ROOT
python - <<PY | tee $OUT/summary.txt
import csv, glob, os
from collections import defaultdict
out = "$OUT"
for d in sorted(glob.glob(out + "/*_p1")):
    name = os.path.basename(d)[:-3]
    acc = defaultdict(lambda: defaultdict(list))
    for p in (d, d[:-1] + "2"):
        fs = glob.glob(p + "/**/*counter_collection.csv", recursive=True)
        if not fs: continue
        for row in csv.DictReader(open(fs[0])):
            n = row["Kernel_Name"]
            if "gemm_kernel" in n:
                key = (n[n.index("gemm_kernel"):][:48], row.get("Grid_Size", ""))
                acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("##", name)
    for (n, grid), c in acc.items():
        m = defaultdict(float, {k: sum(v) / len(v) for k, v in c.items()})
        simd_cyc = m["GRBM_GUI_ACTIVE"] / 8 * 1024            # SIMD-cycles of the kernel
        wc = max(m["SQ_WAVE_CYCLES"], 1.0)
        print(f"{n} grid={grid} gui={m['GRBM_GUI_ACTIVE']:.4g} mfma_busy={m['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cyc:.3f} "
              f"wait_any/wc={m['SQ_WAIT_ANY'] / wc:.3f} active_any/wc={m['SQ_ACTIVE_INST_ANY'] / wc:.3f} "
              f"valu/wc={m['SQ_ACTIVE_INST_VALU'] / wc:.3f} lds/wc={m['SQ_ACTIVE_INST_LDS'] / wc:.3f} vmem/wc={m['SQ_ACTIVE_INST_VMEM'] / wc:.3f} "
              f"insts_valu={m['SQ_INSTS_VALU']:.4g} insts_lds={m['SQ_INSTS_LDS']:.4g} lds_conf/idx={m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1):.3f}")
PY
find $OUT -name "*.csv" -size +5M -delete
