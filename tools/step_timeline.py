"""Ordered kernel timeline of ONE training step from a rocprofv3 --kernel-trace CSV.

    python tools/step_timeline.py <dir-with-*kernel_trace.csv> [marker-substring=knn_wave_kernel]

The step is delimited by consecutive occurrences of the marker kernel (the kNN launch that opens every
step); prints each launch with its duration and the idle gap before it, then totals (busy, gaps)."""
import csv, glob, os, sys

def main():
    d = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "knn_wave_kernel"
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no kernel_trace.csv under " + d
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    assert len(marks) >= 3, "marker not found often enough"
    a, b = marks[-3], marks[-2]            # a full step in steady state (not the last: roofline loop follows)
    step = rows[a:b]
    busy = gaps = 0
    prev_end = step[0][0]
    for s, e, n in step:
        gap = s - prev_end
        gaps += max(gap, 0); busy += e - s
        print(f"{(e - s) / 1e3:8.2f} us  gap {gap / 1e3:7.2f}  {n[:110]}")
        prev_end = max(prev_end, e)
    span = rows[b][0] - step[0][0]
    print(f"# {len(step)} launches, busy {busy / 1e6:.3f} ms, gaps {gaps / 1e6:.3f} ms, span to next step {span / 1e6:.3f} ms")

main()
