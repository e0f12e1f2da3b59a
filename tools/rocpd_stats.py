#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (ROCm 7.2 default output of
`rocprofv3 --kernel-trace --stats`) as a per-kernel table: calls, total/avg/min/max duration.
Usage: python tools/rocpd_stats.py <results.db> [out.csv]"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    name = name.replace("void ", "").replace("pk::", "").replace("pq::", "")
    return name.strip()


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namec = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {namec}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0, 10 ** 18, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    table = sorted(agg.items(), key=lambda kv: -kv[1][1])
    out = [("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct")]
    for k, (n, t, mn, mx) in table:
        out.append((k, n, round(t / 1e3, 1), round(t / n / 1e3, 2), round(mn / 1e3, 2), round(mx / 1e3, 2),
                    round(100.0 * t / total, 2)))
    w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    w.writerows(out)


if __name__ == "__main__":
    main()
