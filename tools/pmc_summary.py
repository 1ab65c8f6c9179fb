"""Per-kernel averages of a rocprofv3 --pmc run (counter_collection csv)."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
if not files:
    print("no counter_collection.csv in", d)
    sys.exit(0)
acc = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open(files[0])):
    name = row.get("Kernel_Name", "")
    if ("fwd_kernel" not in name and "_T_kernel" not in name and "tile_unit_kernel" not in name and "tileT_kernel" not in name
            and "knn_max_bwd" not in name):
        continue
    import re
    m = re.search(r"tile_(?:fwd|unit)_kernel<\d+, \d+, dctile::(\w+)", name)
    mt = re.search(r"tileT_kernel<\d+, \d+, (?:dctileT::)?(\w+)", name)
    mg = re.search(r"ell_T_kernel<\d+, dcell::(\w+)", name)
    short = (("tile_" + m.group(1)) if m else ("tileT_" + mt.group(1)) if mt else ("ell_T_" + mg.group(1)) if mg
             else name.split("(")[0].split("::")[-1])
    acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for kname, ctrs in acc.items():
    print(kname, {c: round(sum(v) / len(v), 1) for c, v in ctrs.items()}, "dispatches", len(next(iter(ctrs.values()))))
