#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${PMC_TAG:-r02o_pmc_gemm}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -o p -- python $GRAFT_REPO_ROOT/tools/gemm_pmc.py > $OUT/p1.log 2>&1; echo rc=$?
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/p2 -o p -- python $GRAFT_REPO_ROOT/tools/gemm_pmc.py > $OUT/p2.log 2>&1; echo rc=$?
python - <<'PY'
import csv, glob, os
from collections import defaultdict
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/${PMC_TAG:-r02o_pmc_gemm}"
for p in ("p1", "p2"):
    f = glob.glob(out + f"/{p}/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        n = row["Kernel_Name"]
        if "gemm_kernel" in n or n.startswith("Cijk"):
            acc[n[:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, c in acc.items():
        print(p, k, " ".join(f"{n}={sum(v)/len(v):.5g}" for n, v in c.items()))
PY
