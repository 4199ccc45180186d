#!/usr/bin/env python3
"""Kernel launch sequence of one steady-state step from a rocprofv3 --kernel-trace CSV (run-length collapsed).

usage: python tools/seq_report.py <kernel_trace.csv> [anchor]"""
import csv
import re
import sys


def main(path, anchor="ingest_kernel"):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("gims::", "")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[2].startswith(anchor)]
    rows = rows[marks[-2]:marks[-1]]
    out, prev, n, dur = [], None, 0, 0.0
    for s, e, name in rows:
        name = re.sub(r"<.*", "", name)
        if name == prev:
            n += 1
            dur += (e - s) / 1e3
        else:
            if prev:
                out.append((prev, n, dur))
            prev, n, dur = name, 1, (e - s) / 1e3
    out.append((prev, n, dur))
    for name, n, dur in out:
        print(f"{n:4d} x {name:40s} {dur:9.1f} us")


if __name__ == "__main__":
    main(*sys.argv[1:])
