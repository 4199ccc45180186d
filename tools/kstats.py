#!/usr/bin/env python3
"""Print per-kernel rows of a rocprofv3 --stats kernel csv (name filter optional):  python tools/kstats.py <dir or csv> [substring ...]"""
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    keys = sys.argv[2:]
    if os.path.isdir(path):
        c = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))
        if not c:
            sys.exit(f"no *kernel_stats.csv under {path}")
        path = c[-1]
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        if not keys or any(k in r["Name"] for k in keys):
            print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s}  avg {float(r['AverageNs']) / 1e3:9.1f} us  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms  {float(r['Percentage']):5.1f} %")


if __name__ == "__main__":
    main()
