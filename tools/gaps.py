#!/usr/bin/env python
"""List the idle gaps on the GPU timeline from a rocprofv3 --kernel-trace csv
(kernel before the gap, kernel after, gap length).  Usage: gaps.py kernel_trace.csv [min_ms]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
thr = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 2e6
busy = sum(e - s for s, e, _ in rows)
print("kernels %d  span %.1f ms  busy %.1f ms" % (len(rows), (rows[-1][1] - rows[0][0]) / 1e6, busy / 1e6))
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    if s1 - e0 > thr:
        print("gap %8.2f ms at t=%9.1f ms  after %-60s before %s" % ((s1 - e0) / 1e6, (e0 - rows[0][0]) / 1e6, n0, n1))
