#!/usr/bin/env python3
"""Per-tick PMC counters of one GN-20 batch: lists the counter values of every k_nn_pruned / k_nn_compact / k_linearize dispatch in launch order.
usage: per_tick_pmc.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select dispatch_id, kernel_name, grid_size_x, grid_size_y, counter_name, value from counters_collection "
                  "where kernel_name like '%k_nn_pruned%' or kernel_name like '%k_nn_compact%' or kernel_name like '%k_linearize%' or kernel_name like '%knn_cov%' order by dispatch_id").fetchall()
by = {}
for d, k, gx, gy, c, v in rows:
    by.setdefault(d, [k.split("(")[0].replace("void apd::", ""), gx, gy, {}])[3][c] = v
last = None
n = 0
for d in sorted(by):
    k, gx, gy, cs = by[d]
    if gy < 8:
        continue
    waves = cs.get("SQ_WAVES", 1.0)
    print(f"{d:6d} {k:24s} grid=({gx},{gy}) " + " ".join(f"{c}={v:.0f}" for c, v in sorted(cs.items())) +
          f"  VALU/wave={cs.get('SQ_INSTS_VALU', 0) / waves:.0f} SALU/wave={cs.get('SQ_INSTS_SALU', 0) / waves:.0f}")
    n += 1
    if n > 140:
        break
