#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel table: calls, total/avg/min/max us, %."""
import re
import sqlite3
import sys


def main(path, top=40):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else scol[-1])
    q = (f"select s.{name_col}, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
         f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc")
    rows = cur.execute(q).fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"{'kernel':70s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for name, n, t, mn, mx in rows[:top]:
        name = re.sub(r"\(.*", "", name)[:70]
        print(f"{name:70s} {n:7d} {t / 1e3:12.1f} {t / n / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100.0 * t / tot:6.2f}")
    print(f"{'TOTAL':70s} {sum(r[1] for r in rows):7d} {tot / 1e3:12.1f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
