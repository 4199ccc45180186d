#!/usr/bin/env python3
"""One steady-state training step from a rocprofv3 --kernel-trace CSV: span, union of busy time, and per kernel the summed duration and the
EXCLUSIVE time (intervals in which no other kernel runs) -- with the parameter-gradient jobs on a side stream the sum of durations overstates
what a kernel costs the step.

usage: python tools/train_timeline.py <kernel_trace.csv> [anchor=adam_kernel]"""
import csv
import re
import sys
from collections import defaultdict


def main(path, anchor="adam_kernel"):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("gims::", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[2].startswith(anchor)]
    # the anchor is launched several times per step (parameter chunks): a step ends with the LAST anchor of a run of anchors
    ends = [i for k, i in enumerate(marks) if k + 1 == len(marks) or marks[k + 1] - i > 50]
    rows = rows[ends[-2] + 1:ends[-1] + 1]
    span = rows[-1][1] - rows[0][0]
    ev = []
    for k, (s, e, n, q) in enumerate(rows):
        ev.append((s, 1, k))
        ev.append((e, 0, k))
    ev.sort()
    live, last, union = set(), None, 0
    excl = defaultdict(float)
    for t, kind, k in ev:
        if live and last is not None:
            union += t - last
            if len(live) == 1:
                excl[rows[next(iter(live))][2]] += t - last
        if kind:
            live.add(k)
        else:
            live.discard(k)
        last = t
    tot = defaultdict(lambda: [0, 0.0])
    queues = defaultdict(float)
    for s, e, n, q in rows:
        tot[n][0] += 1
        tot[n][1] += e - s
        queues[q] += e - s
    print(f"{len(rows)} kernels, span {span / 1e6:.2f} ms, union busy {union / 1e6:.2f} ms, idle {(span - union) / 1e6:.2f} ms, sum {sum(v[1] for v in tot.values()) / 1e6:.2f} ms")
    print("per queue (sum ms):", {q: round(v / 1e6, 2) for q, v in queues.items()})
    for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"{d / 1e6:8.2f} ms  excl {excl[n] / 1e6:7.2f} ms  {c:5d} x {d / c / 1e3:7.1f} us  {n[:100]}")


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]))
