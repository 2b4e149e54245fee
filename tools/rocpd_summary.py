#!/usr/bin/env python
"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite) outputs as text:
  python tools/rocpd_summary.py stats <results.db> [last_ms] per-kernel time table (--kernel-trace --stats)
  python tools/rocpd_summary.py pmc   <results.db> [...]    per-kernel mean counter values (--pmc passes)
  python tools/rocpd_summary.py gaps  <results.db> [n [last_ms]]   the n longest idle stretches between two kernels, with the
                                                            kernels on either side, and the idle time by the kernel that
                                                            follows; last_ms: only the last so many ms of the trace
"""
import sqlite3
import sys
from collections import defaultdict


def short(name, n=70):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name if len(name) <= n else name[:n - 3] + "..."


def stats(path, last_ms=None):
    cur = sqlite3.connect(path).cursor()
    where = ""
    if last_ms is not None:                                  # only the launches of the last so many ms of the trace
        where = " where start >= %d" % (cur.execute("select max(end) from kernels").fetchone()[0] - int(last_ms * 1e6))
    rows = cur.execute("select name, duration, workgroup_x, grid_x, vgpr_count, sgpr_count, lds_size, scratch_size "
                       "from kernels" + where).fetchall()
    agg = defaultdict(list)
    meta = {}
    for name, dur, wg, grid, vg, sg, lds, scr in rows:
        agg[name].append(dur)
        meta[name] = (wg, grid, vg, sg, lds, scr)
    total = sum(sum(v) for v in agg.values())
    n, t0, t1 = cur.execute("select count(*), min(start), max(end) from kernels" + where).fetchone()
    print("# %d kernel launches, GPU busy %.3f ms, first-start to last-end span %.3f ms" % (n, total / 1e6, (t1 - t0) / 1e6))
    print("%-72s %6s %12s %12s %12s %12s %6s | %5s %8s %5s %5s %7s %5s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%", "wg", "grid", "vgpr", "sgpr", "lds", "scr"))
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        wg, grid, vg, sg, lds, scr = meta[name]
        print("%-72s %6d %12.1f %12.1f %12.1f %12.1f %6.2f | %5d %8d %5d %5d %7d %5d" % (
            short(name), len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3,
            100.0 * sum(v) / total, wg, grid, vg, sg, lds, scr))


def gaps(path, n=25, last_ms=None):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    if last_ms is not None:
        t_end = max(r[2] for r in rows)
        rows = [r for r in rows if r[1] >= t_end - last_ms * 1e6]
        busy = sum(r[2] - r[1] for r in rows)
        print("# last %.1f ms of the trace: %d launches, busy %.3f ms" % (last_ms, len(rows), busy / 1e6))
    out, by_next = [], defaultdict(lambda: [0, 0.0])
    last_end, last_name = None, None
    for name, st, en in rows:
        if last_end is not None and st > last_end:
            g = (st - last_end) / 1e3
            out.append((g, last_name, name, (st - rows[0][1]) / 1e6))
            by_next[name][0] += 1
            by_next[name][1] += g
        if last_end is None or en > last_end:
            last_end, last_name = en, name
    total = sum(g for g, _, _, _ in out)
    print("# %d launches, idle between kernels %.3f ms in %d gaps" % (len(rows), total / 1e3, len(out)))
    print("# longest gaps: us | at ms | after kernel -> before kernel")
    for g, a, b, at in sorted(out, key=lambda x: -x[0])[:n]:
        print("%10.1f | %9.2f | %s -> %s" % (g, at, short(a, 50), short(b, 50)))
    print("# idle time by the kernel that follows: total us | gaps | mean us | kernel")
    for name, (c, t) in sorted(by_next.items(), key=lambda kv: -kv[1][1])[:n]:
        print("%10.1f | %5d | %8.1f | %s" % (t, c, t / c, short(name, 70)))


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
        stats(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else None)
    elif sys.argv[1] == "gaps":
        gaps(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 25, float(sys.argv[4]) if len(sys.argv) > 4 else None)
    else:
        pmc(sys.argv[2:])
