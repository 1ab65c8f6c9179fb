#!/bin/bash
# SQ counters of the depth-2 edge-MLP kernels (csrc/edge2.hip) at the ShapeNet shape: two --pmc passes, --kernel-trace only beside them.
#   gpurun -- 'bash tools/pmc_edge2.sh <tag>'
TAG=${1:-pmc_edge2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; ulimit -c 0
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
timeout 200 rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $OUT/p1 -o p -- python $GRAFT_REPO_ROOT/tools/edge2_bench.py > $OUT/p1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $OUT/p2 -o p -- python $GRAFT_REPO_ROOT/tools/edge2_bench.py > $OUT/p2.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY | tee $OUT/summary.txt
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for p in ("$OUT/p1", "$OUT/p2"):
    for f in glob.glob(p + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            n = row["Kernel_Name"]
            if "edge2_" in n:
                acc[n[n.index("edge2_"):][:28]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("# tools/pmc_edge2.sh: csrc/edge2.hip at 16 x 2048 points, k = 20 (E = 655 360 edges); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles")
print("# (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 x 1024 SIMDs); two separate --pmc passes")
for n, c in acc.items():
    m = defaultdict(float, {k: sum(v) / len(v) for k, v in c.items()})
    simd_cyc = m["GRBM_GUI_ACTIVE"] / 8 * 1024
    wc = max(m["SQ_WAVE_CYCLES"], 1.0)
    print(f"{n:28s} gui={m['GRBM_GUI_ACTIVE']:.4g} mfma_busy={m['SQ_VALU_MFMA_BUSY_CYCLES'] / max(simd_cyc, 1):.3f} wait_any/wc={m['SQ_WAIT_ANY'] / wc:.3f} "
          f"active_any/wc={m['SQ_ACTIVE_INST_ANY'] / wc:.3f} valu/wc={m['SQ_ACTIVE_INST_VALU'] / wc:.3f} lds/wc={m['SQ_ACTIVE_INST_LDS'] / wc:.3f} "
          f"vmem/wc={m['SQ_ACTIVE_INST_VMEM'] / wc:.3f} insts_valu={m['SQ_INSTS_VALU']:.4g} insts_lds={m['SQ_INSTS_LDS']:.4g} "
          f"lds_conf/idx={m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1):.3f}")
PY
find $OUT -name "*.csv" -size +5M -delete
