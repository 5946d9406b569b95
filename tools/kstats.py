#!/usr/bin/env python3
"""Prints the per-kernel summary of a `rocprofv3 --kernel-trace --stats --output-format csv` output directory (top entries).
   python tools/kstats.py DIR [N]"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    files = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if not files:
        print("no *kernel_stats.csv under", d)
        return 1
    rows = list(csv.DictReader(open(files[0])))
    for r in rows[:n]:
        print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s} total_ms {float(r["TotalDurationNs"]) / 1e6:10.2f} avg_us {float(r["AverageNs"]) / 1e3:10.1f} {r["Percentage"]}%')
    return 0


if __name__ == "__main__":
    sys.exit(main())
