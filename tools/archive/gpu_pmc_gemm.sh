#!/bin/bash
# SQ counters of the forward GEMM on model shapes, own kernel vs vendor library: tools/gpu_pmc_gemm.sh <tag>
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_gemm}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -o p -- python $GRAFT_REPO_ROOT/tools/gemm_pmc.py > $OUT/p1.log 2>&1; echo rc=$?
python - <<PY
import csv, glob
from collections import defaultdict
f = glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open(f)):
    n = row["Kernel_Name"]
    if "gemm_kernel" in n or n.startswith("Cijk"):
        acc[(n[:70], row.get("Grid_Size", row.get("Grid_Size_X", "")))][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("$OUT/summary.txt", "w") as fh:
    for (n, grid), c in acc.items():
        m = {k: sum(v) / len(v) for k, v in c.items()}
        line = f"{n} grid={grid} mfma_busy={m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f} wait_any/wave_cycles={m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f} wait_inst/wave_cycles={m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f} GRBM_GUI_ACTIVE={m['GRBM_GUI_ACTIVE']:.4g}"
        print(line); fh.write(line + "\n")
PY
