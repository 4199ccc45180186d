#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files into HBM bytes per launch per kernel.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced read stream, so it is doubled; WRITE_SIZE is
taken as reported (uncalibrated).  Separate passes (TCC has 4 slots: FETCH_SIZE uses 3, WRITE_SIZE 2)."""
import csv
import json
import re
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("gims::", "")
            k = re.sub(r"<.*", "", k)
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in acc.items()}


def main(fetch_csv, write_csv, key, out_json):
    fe, wr = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fe) | set(wr)):
        f_kib = fe.get(k, (0, 0.0))[1]
        w_kib = wr.get(k, (0, 0.0))[1]
        res[k] = {"launches_profiled": fe.get(k, (0, 0))[0], "fetch_kib_raw": f_kib, "write_kib_raw": w_kib,
                  "hbm_bytes_per_launch": (2.0 * f_kib + w_kib) * 1024.0}
    try:
        allr = json.load(open(out_json))
    except Exception:
        allr = {}
    allr[key] = res
    import os
    if os.environ.get("GIMS_HEAD"):          # the commit the counters were taken on (tools/refresh_profiles.sh <tag> <head>): a stale table is visible in the bench line
        allr["_meta"] = {"head": os.environ["GIMS_HEAD"]}
    json.dump(allr, open(out_json, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print(f"{k:32s} n={v['launches_profiled']:5d} fetch(raw KiB)={v['fetch_kib_raw']:12.1f} write(KiB)={v['write_kib_raw']:12.1f} "
              f"HBM bytes/launch={v['hbm_bytes_per_launch'] / 1e6:10.2f} MB")


if __name__ == "__main__":
    main(*sys.argv[1:5])
