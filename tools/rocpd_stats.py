#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (ROCm 7.2 default output of
`rocprofv3 --kernel-trace --stats`) as a per-kernel table: calls, total/avg/min/max duration - and, since round 6, the
kernel's share of the TIMELINE: avg_step_us = mean of end(k) - end(previous kernel) (gaps > 20 us are host time and not
counted), i.e. the duration plus the boundary in front of it.  Measured on this stack the two columns agree within 0.1-0.5 us:
traced kernels run strictly one after the other, each carrying ~1.1 us of tracing overhead (a traced decode_mode 0 step adds
up to 3.47 ms of kernel time against 2.9 ms un-traced) - read trivial kernels' `avg_us` with that in mind.
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
    rows = db.execute(f"select {namec}, start, end from kernels order by end").fetchall()
    agg = {}
    prev_end = None
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0, 10 ** 18, 0, 0, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
        # timeline share: from the previous kernel's end (or this kernel's own start, if the stream was idle in between:
        # gaps > 20 us are host time, not counted) to this kernel's end
        if prev_end is not None and s - prev_end < 20000:
            a[4] += e - max(prev_end, min(s, prev_end)); a[5] += 1
        else:
            a[4] += d; a[5] += 1
        prev_end = e
    total = sum(a[1] for a in agg.values()) or 1
    table = sorted(agg.items(), key=lambda kv: -kv[1][1])
    out = [("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "avg_step_us")]
    for k, (n, t, mn, mx, st, sn) in table:
        out.append((k, n, round(t / 1e3, 1), round(t / n / 1e3, 2), round(mn / 1e3, 2), round(mx / 1e3, 2),
                    round(100.0 * t / total, 2), round(st / max(sn, 1) / 1e3, 2)))
    w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    w.writerows(out)


if __name__ == "__main__":
    main()
