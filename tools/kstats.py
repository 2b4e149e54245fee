#!/usr/bin/env python
"""Compact view of a rocprofv3 kernel_stats.csv: kstats.py file.csv [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total GPU busy %.1f ms over %d kernel names" % (tot / 1e6, len(rows)))
for r in rows[:n]:
    print("%-72s %6s %9.2f ms %9.1f us %6.2f%%" % (r["Name"][:72], r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                                   float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
