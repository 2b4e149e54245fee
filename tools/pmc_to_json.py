#!/usr/bin/env python
"""profiles/pmc_traffic.json from two rocprofv3 --pmc result databases (FETCH_SIZE pass, WRITE_SIZE pass):
   python tools/pmc_to_json.py <fetch.db> <write.db> <replicas> <frames> > profiles/pmc_traffic.json
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE counts 64 B per 128 B request on gfx950
for wide streaming reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as reported (uncalibrated)."""
import json
import sqlite3
import sys


def mean(db, counter, kernel):
    cur = sqlite3.connect(db).cursor()
    v = [float(r[0]) for r in cur.execute(
        "select value from counters_collection where counter_name=? and kernel_name like ?", (counter, "%" + kernel + "%"))]
    return sum(v) / len(v) if v else None


fetch_db, write_db, R, T = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
out = {}
for k in ("traj_adj_ring_kernel", "traj_fwd_ring_kernel", "traj_adj_kernel", "traj_fwd_kernel", "rdf_fwd_fine_kernel", "rdf_fwd_half_kernel", "rdf_fwd_lane_kernel", "rdf_fwd_block8_kernel",
          "rdf_bwd_fine_kernel", "rdf_bwd_kernel"):
    f, w = mean(fetch_db, "FETCH_SIZE", k), mean(write_db, "WRITE_SIZE", k)
    if f is None or w is None:
        continue
    out[k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "replicas": R, "frames": T,
              "hbm_bytes_per_launch": (2 * f + w) * 1024, "hbm_bytes_per_replica": (2 * f + w) * 1024 / R}
print(json.dumps(out, indent=1))
