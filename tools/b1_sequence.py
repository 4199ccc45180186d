#!/usr/bin/env python3
"""Ordered kernel sequence of ONE steady-state forward() call from a rocprofv3 --kernel-trace CSV of tools/b1_loop.py: start offset, duration and the
gap in front of every kernel (us).   python tools/b1_sequence.py <kernel_trace.csv> [min_gap_us]
(with min_gap_us: only the kernels that start after an idle gap of at least that much -- where a batched step leaves the device waiting)"""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("gims::", "")))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith("ingest_kernel")]
nxt = rows[marks[-1]][0]                   # start of the next call's first kernel
rows = rows[marks[-2]:marks[-1]]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else None
t0, end = rows[0][0], rows[0][0]
print(f"{len(rows)} kernels, span {(rows[-1][1] - t0) / 1e3:.1f} us, busy {sum(e - s for s, e, _ in rows) / 1e3:.1f} us")
idle = 0.0
for s, e, n in rows:
    idle += max(0, s - end) / 1e3
    if min_gap is None or (s - end) / 1e3 >= min_gap:
        print(f"{(s - t0) / 1e3:9.1f}  +{(s - end) / 1e3:6.1f}  {(e - s) / 1e3:7.1f}  {n[:100]}")
    end = max(end, e)
print(f"idle inside the span: {idle:.1f} us; until the next call's first kernel: +{max(0, nxt - end) / 1e3:.1f} us (call to call {(nxt - t0) / 1e3:.1f} us)")
