#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -o p -- python $GRAFT_REPO_ROOT/tools/gemm_driver.py > $OUT/p1.log 2>&1; echo rc=$?
python - <<'PY'
import csv, glob, re, os
from collections import defaultdict
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_gemm"
f = glob.glob(out + "/p1/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open(f[0])):
    n = row["Kernel_Name"]
    if "gemm_tn_kernel" in n or n.startswith("Cijk"):
        key = (n[:60], row["Grid_Size_X"] if "Grid_Size_X" in row else "")
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
kt = glob.glob(out + "/p1/**/*kernel_trace.csv", recursive=True)
dur = defaultdict(list)
for row in csv.DictReader(open(kt[0])):
    n = row["Kernel_Name"]
    if "gemm_tn_kernel" in n or n.startswith("Cijk"):
        dur[(n[:60], row.get("Grid_Size_X", ""))].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
with open(out + "/summary.txt", "w") as fh:
    for k, c in acc.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        us = sum(dur[k]) / max(len(dur[k]), 1)
        line = f"{k[0]} grid={k[1]} us={us:.1f} " + " ".join(f"{n}={v:.4g}" for n, v in m.items())
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
            # MFMA pipe busy cycles summed over SIMDs / (4 SIMDs x 256 CUs x active cycles)
            line += f"  mfma_busy_frac={m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] * 1024):.3f}"
        print(line); fh.write(line + "\n")
PY
