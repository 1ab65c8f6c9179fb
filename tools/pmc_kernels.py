"""Per-kernel averages of a rocprofv3 --pmc run, keyed by the full (template-qualified) kernel name.
usage: python tools/pmc_kernels.py <rocprof output dir> [substring filter]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
if not files:
    print("no counter_collection.csv in", d)
    sys.exit(0)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if flt and flt not in name:
            continue
        short = re.sub(r"\(.*", "", name).replace("void ", "")
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for kname, ctrs in acc.items():
    print(kname, {c: round(sum(v) / len(v), 1) for c, v in ctrs.items()}, "dispatches", len(next(iter(ctrs.values()))))
