"""Launch-by-launch comparison of two step timelines (tools/step_timeline.py outputs of two builds of the library
running the same step): same launch sequence, so launches are aligned by position.
    python tools/timeline_diff.py a.txt b.txt  -> per launch: us(a) us(b) delta name; totals per kernel family"""
import re
import sys
from collections import defaultdict


def load(path):
    rows = []
    for ln in open(path):
        m = re.match(r"\s*([\d.]+) us  gap\s+(-?[\d.]+)\s+(.*)", ln)
        if m:
            rows.append((float(m.group(1)), float(m.group(2)), m.group(3).strip()))
    return rows


a, b = load(sys.argv[1]), load(sys.argv[2])
print(f"# {len(a)} vs {len(b)} launches")
fam = defaultdict(lambda: [0.0, 0.0, 0])
if len(a) == len(b) and all(x[2][:40] == y[2][:40] for x, y in zip(a, b)):
    for i, (x, y) in enumerate(zip(a, b)):
        print(f"{i:4d} {x[0]:8.2f} {y[0]:8.2f} {y[0] - x[0]:+7.2f}  gaps {x[1]:6.2f} {y[1]:6.2f}  {x[2][:90]}")
        key = re.sub(r"<.*", "", x[2])[:50]
        fam[key][0] += x[0]; fam[key][1] += y[0]; fam[key][2] += 1
else:
    for rows, col in ((a, 0), (b, 1)):
        for t, _, n in rows:
            key = re.sub(r"<.*", "", n)[:50]
            fam[key][col] += t
            fam[key][2] += col == 0
print("# per kernel name: sum us (a), sum us (b), delta, launches")
for k, (ta, tb, n) in sorted(fam.items(), key=lambda kv: kv[1][1] - kv[1][0]):
    print(f"{ta:9.1f} {tb:9.1f} {tb - ta:+8.1f} {n:4d}  {k}")
print(f"# total busy {sum(x[0] for x in a):.1f} vs {sum(x[0] for x in b):.1f} us; gaps {sum(max(x[1],0) for x in a):.1f} vs {sum(max(x[1],0) for x in b):.1f}")
