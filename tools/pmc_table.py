#!/usr/bin/env python3
"""Per-kernel averages of arbitrary PMC counters from a rocprofv3 counter_collection.csv."""
import csv, re, sys
from collections import defaultdict

def main(path, *kernels):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    dur = defaultdict(lambda: [0, 0.0])
    seen = set()
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("gims::", "")
        if kernels and not any(x in k for x in kernels):
            continue
        a = acc[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"],)
        if key not in seen:
            seen.add(key)
            dur[k][0] += 1; dur[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k in acc:
        print(f"== {k}  launches={dur[k][0]}  avg_us={dur[k][1] / max(1, dur[k][0]):.1f}")
        for c, (n, v) in sorted(acc[k].items()):
            print(f"   {c:28s} {v / n:16.1f}")

if __name__ == "__main__":
    main(*sys.argv[1:])
