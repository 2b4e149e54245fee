#!/usr/bin/env python
"""profiles/pmc_<workload>.json from rocprofv3 result databases of ONE bench workload (rocpd sqlite, ROCm 7.2):

    python tools/pmc_collect.py <workload> <stats.db> <pmc1.db> [<pmc2.db> ...] > profiles/pmc_<workload>.json

<stats.db> is a --kernel-trace --stats run (durations without counter overhead); every other database is one --pmc pass
(FETCH_SIZE, WRITE_SIZE in passes of their own, as MI355X_MICROARCH.md prescribes; issue-side counters grouped).  Per
kernel: calls, mean duration, share of the GPU-busy time, mean counter values and what is derived from them:

    hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024      FETCH_SIZE counts 64 B per 128 B request on gfx950
    valu_busy  = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * duration * 2.4 GHz)
    mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * duration * 2.4 GHz)     (the counter sums over the chip's SIMDs)
    wait_frac  = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                    share of wave-cycles spent waiting on any instruction

The file is stamped with `sources`: sha256[:16] of every file under mdgrad_amd/csrc; bench.py refuses counters whose
kernel's source file changed since (profiles/README.md)."""
import glob
import hashlib
import json
import os
import sqlite3
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hashes():
    out = {}
    for p in sorted(glob.glob(os.path.join(ROOT, "mdgrad_amd", "csrc", "*"))):
        if os.path.isfile(p):
            out[os.path.basename(p)] = hashlib.sha256(open(p, "rb").read()).hexdigest()[:16]
    return out


def clean(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


def main():
    workload, stats_db, pmc_dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
    cur = sqlite3.connect(stats_db).cursor()
    rows = cur.execute("select name, duration, workgroup_x, grid_x, vgpr_count, lds_size, scratch_size from kernels").fetchall()
    dur, meta = defaultdict(list), {}
    for name, d, wg, grid, vg, lds, scr in rows:
        k = clean(name)
        dur[k].append(float(d))
        meta[k] = {"workgroup": wg, "grid": grid, "vgpr": vg, "lds": lds, "scratch": scr}
    n, t0, t1 = cur.execute("select count(*), min(start), max(end) from kernels").fetchone()
    busy = sum(sum(v) for v in dur.values())
    kernels = {}
    for k, v in dur.items():
        if sum(v) < 0.002 * busy:
            continue
        kernels[k] = dict(meta[k], calls=len(v), avg_us=sum(v) / len(v) / 1e3, share=sum(v) / busy, counters={})
    for db in pmc_dbs:
        c = sqlite3.connect(db).cursor()
        agg = defaultdict(list)
        for name, cn, val, d in c.execute("select kernel_name, counter_name, value, duration from counters_collection"):
            agg[(clean(name), cn)].append((float(val), float(d)))
        for (k, cn), v in agg.items():
            if k in kernels:
                kernels[k]["counters"][cn] = sum(x[0] for x in v) / len(v)
                kernels[k].setdefault("avg_us_under_pmc", {})[cn] = sum(x[1] for x in v) / len(v) / 1e3
    for k, r in kernels.items():
        c, sec = r["counters"], r["avg_us"] * 1e-6
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            r["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            r["hbm_gbs"] = r["hbm_bytes_per_launch"] / sec / 1e9
        # issue counters are rated against the duration of the pass that collected them
        for cn, key, scale in (("SQ_ACTIVE_INST_VALU", "valu_busy", 4.0), ("SQ_VALU_MFMA_BUSY_CYCLES", "mfma_busy", 1.0)):
            if cn in c:
                s = r["avg_us_under_pmc"][cn] * 1e-6
                r[key] = c[cn] * scale / (1024.0 * s * 2.4e9)
        if "SQ_WAIT_INST_ANY" in c and c.get("SQ_WAVE_CYCLES"):
            r["wait_frac"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
        r.pop("avg_us_under_pmc", None)
    out = {"workload": workload, "launches": n, "gpu_busy_ms": busy / 1e6, "span_ms": (t1 - t0) / 1e6,
           "busy_over_span": busy / max(1.0, (t1 - t0)), "clock_assumed_ghz": 2.4, "sources": source_hashes(),
           "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["share"]))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
