#!/usr/bin/env python3
"""Idle-time report from a rocprofv3 --kernel-trace CSV: where the GPU waits between kernels.

usage: python tools/gap_report.py <kernel_trace.csv> [min_gap_us]"""
import csv
import re
import sys
from collections import defaultdict


def main(path, min_gap=15.0, anchor="ingest_kernel"):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("gims::", "")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[2].startswith(anchor)]
    if len(marks) >= 3:                       # one steady-state step: between the last two anchor kernels
        rows = rows[marks[-2]:marks[-1]]
        print(f"window: one step between the last two '{anchor}' launches")
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    print(f"{len(rows)} kernels, span {span / 1e6:.2f} ms, sum of durations {busy / 1e6:.2f} ms")
    gaps = defaultdict(lambda: [0, 0.0])
    small = 0.0
    end = rows[0][1]
    prev = rows[0][2]
    for s, e, n in rows[1:]:
        g = (s - end) / 1e3
        if g > min_gap:
            k = f"{prev} -> {n}"
            gaps[k][0] += 1
            gaps[k][1] += g
        elif g > 0:
            small += g
        if e > end:
            end, prev = e, n
    print(f"gaps <= {min_gap} us: total {small / 1e3:.2f} ms")
    by_kernel = defaultdict(lambda: [0, 0.0])
    for s0, e0, n in rows:
        by_kernel[n][0] += 1
        by_kernel[n][1] += (e0 - s0) / 1e3
    print("top kernels in the window:")
    for k, (c, t) in sorted(by_kernel.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"{t / 1e3:8.3f} ms  {c:4d}x  {k[:90]}")
    print("largest gaps:")
    for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{t / 1e3:8.3f} ms  {c:4d}x  avg {t / c:8.1f} us   {k[:150]}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 15.0)
