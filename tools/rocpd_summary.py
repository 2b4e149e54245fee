#!/usr/bin/env python
"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite) outputs as text:
  python tools/rocpd_summary.py stats <results.db>          per-kernel time table (--kernel-trace --stats)
  python tools/rocpd_summary.py pmc   <results.db> [...]    per-kernel mean counter values (--pmc passes)
"""
import sqlite3
import sys
from collections import defaultdict


def short(name, n=70):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name if len(name) <= n else name[:n - 3] + "..."


def stats(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, duration, workgroup_x, grid_x, vgpr_count, sgpr_count, lds_size, scratch_size "
                       "from kernels").fetchall()
    agg = defaultdict(list)
    meta = {}
    for name, dur, wg, grid, vg, sg, lds, scr in rows:
        agg[name].append(dur)
        meta[name] = (wg, grid, vg, sg, lds, scr)
    total = sum(sum(v) for v in agg.values())
    n, t0, t1 = cur.execute("select count(*), min(start), max(end) from kernels").fetchone()
    print("# %d kernel launches, GPU busy %.3f ms, first-start to last-end span %.3f ms" % (n, total / 1e6, (t1 - t0) / 1e6))
    print("%-72s %6s %12s %12s %12s %12s %6s | %5s %8s %5s %5s %7s %5s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%", "wg", "grid", "vgpr", "sgpr", "lds", "scr"))
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        wg, grid, vg, sg, lds, scr = meta[name]
        print("%-72s %6d %12.1f %12.1f %12.1f %12.1f %6.2f | %5d %8d %5d %5d %7d %5d" % (
            short(name), len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3,
            100.0 * sum(v) / total, wg, grid, vg, sg, lds, scr))


def pmc(paths):
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        rows = cur.execute("select kernel_name, counter_name, value, duration from counters_collection").fetchall()
        agg = defaultdict(list)
        for name, cn, val, dur in rows:
            agg[(name, cn)].append((float(val), dur))
        print("# %s" % path)
        print("%-72s %-12s %6s %16s %12s" % ("kernel", "counter", "calls", "mean_value", "mean_us"))
        for (name, cn), v in sorted(agg.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
            print("%-72s %-12s %6d %16.2f %12.1f" % (short(name), cn, len(v), sum(x[0] for x in v) / len(v),
                                                    sum(x[1] for x in v) / len(v) / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
