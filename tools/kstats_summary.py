#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --stats run, per call of the probed function: python3 tools/kstats_summary.py <dir> <calls> [rows]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
calls = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
print("GPU busy per call %.0f us" % (sum(float(r["TotalDurationNs"]) for r in rows) / calls / 1e3))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 26]:
    print("%-72s launches/call %6.1f avg %8.1f us  per call %8.1f us" % (r["Name"][:72], int(r["Calls"]) / calls, float(r["AverageNs"]) / 1e3,
                                                                         float(r["TotalDurationNs"]) / calls / 1e3))
