"""Condense a rocprofv3 --kernel-trace --stats output directory into a per-kernel table."""
import csv
import glob
import os
import sys


def main(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        print("no kernel_stats.csv under", d)
        return
    rows = list(csv.DictReader(open(files[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# {files[0]}\n# total kernel time {tot / 1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} launches")
    print(f"{'kernel':80s} {'calls':>7s} {'avg_us':>9s} {'total_ms':>9s} {'pct':>6s}")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:60]:
        print(f"{r['Name'][:80]:80s} {r['Calls']:>7s} {float(r['AverageNs']) / 1e3:9.2f} "
              f"{float(r['TotalDurationNs']) / 1e6:9.3f} {float(r['Percentage']):6.2f}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
