#!/usr/bin/env python3
"""Median kernel durations of tools/gemm_f32_probe.py from a rocprofv3 kernel trace (argument: the trace directory)."""
import csv
import glob
import statistics
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))
       for r in csv.DictReader(open(f)) if "gemm_split" in r["Kernel_Name"] or "splitk" in r["Kernel_Name"]]
chunks, cur, cnt = [], [], 0
for s in seq:
    if "splitk" not in s[0]:
        if cnt == 33:
            chunks.append(cur)
            cur, cnt = [], 0
        cnt += 1
    cur.append(s)
chunks.append(cur)
for c in chunks:
    gm = [d for n, d, *_ in c if "splitk" not in n][3:]
    sk = [d for n, d, *_ in c if "splitk" in n][3:]
    name = c[0][0]
    print(name[name.index("<"):name.index(">") + 1], "gemm %.1f us" % statistics.median(gm), ("+ splitk fold %.1f us" % statistics.median(sk)) if sk else "",
          "grid", c[0][2], c[0][3], c[0][4])
