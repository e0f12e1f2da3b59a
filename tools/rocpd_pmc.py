#!/usr/bin/env python3
"""Per-kernel average of the PMC counters in a rocprofv3 rocpd sqlite database
(`rocprofv3 --pmc <COUNTER> --kernel-trace`).  Usage: python tools/rocpd_pmc.py <results.db> [out.csv]"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("pk::", "").replace("pq::", "").strip()


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    kname = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "kernel" in c][0]
    cname = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    vname = "value" if "value" in cols else [c for c in cols if "value" in c][0]
    agg = {}
    # --by-grid: one row per (kernel, grid size) - tells the shapes of one kernel apart (qkv / o_proj / lm_head of a decode step)
    gcols = [c for c in cols if c in ("grid_size", "grid_size_x", "grid_x")] if "--by-grid" in sys.argv else []
    if "--by-grid" in sys.argv:
        sys.argv.remove("--by-grid")
        if not gcols:
            print("no grid column among", cols, file=sys.stderr)
    gsel = f", {gcols[0]}" if gcols else ""
    for row in db.execute(f"select {kname}, {cname}, {vname}{gsel} from counters_collection"):
        k, c, v = row[:3]
        if gcols:
            k = f"{k} grid={row[3]}"
        a = agg.setdefault((short(k) + (f" grid={row[3]}" if gcols else ""), c), [0, 0.0])
        a[0] += 1
        a[1] += float(v)
    out = [("kernel", "counter", "dispatches", "avg_value", "total_value")]
    for (k, c), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append((k, c, n, round(t / n, 2), round(t, 1)))
    w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    w.writerows(out)


if __name__ == "__main__":
    main()
