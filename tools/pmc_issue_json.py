#!/usr/bin/env python
"""profiles/pmc_issue.json from a rocprofv3 --pmc pass holding SQ_ACTIVE_INST_VALU: per headline kernel the SIMD
VALU-busy fraction = SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (1024 SIMDs x kernel duration x 2.4 GHz).  The clock is
the nominal 2.4 GHz (the counters carry no clock), so a kernel that runs below it reads slightly low."""
import json
import sqlite3
import sys
from collections import defaultdict

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select kernel_name, counter_name, value, duration from counters_collection").fetchall()
agg = defaultdict(list)
for name, cn, val, dur in rows:
    clean = name.replace("void ", "").replace("(anonymous namespace)::", "")
    agg[(clean.split("<")[0].split("(")[0], cn)].append((float(val), float(dur)))
out = {}
for (name, cn), v in agg.items():
    if cn != "SQ_ACTIVE_INST_VALU":
        continue
    short = name
    val = sum(x[0] for x in v) / len(v)
    dur_s = sum(x[1] for x in v) / len(v) * 1e-9
    if dur_s < 1e-4:
        continue
    out[short] = {"valu_busy": val * 4.0 / (1024.0 * dur_s * 2.4e9), "kernel_ms_profiled": dur_s * 1e3,
                  "SQ_ACTIVE_INST_VALU": val, "clock_assumed_ghz": 2.4}
print(json.dumps(out, indent=1))
