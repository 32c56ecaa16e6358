#!/usr/bin/env python3
"""Timeline of the pooled LM ticks in a rocprofv3 kernel trace: per kernel of the last `ms` milliseconds its start (us from the
window's begin), duration, the gap to the previous kernel on the same stream/queue, and grid y (pairs covered).
usage: rocpd_pool.py results.db [ms=3] [max_rows=400]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
max_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 400
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"select name, start, end, grid_y, workgroup_x, grid_x, {qcol or '0'} from kernels order by start").fetchall()
t_end = rows[-1][2]
rows = [r for r in rows if r[1] >= t_end - ms * 1e6]
t0 = rows[0][1]
last_end = {}
agg = {}
for name, s, e, gy, wx, gx, q in rows[:max_rows]:
    short = name.split("(")[0].replace("void ", "").replace("apd::", "")
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  q{q}  y={gy:<4} x={gx // max(1, wx):<4} {short}")
for name, s, e, gy, wx, gx, q in rows:
    short = name.split("(")[0].replace("void ", "").replace("apd::", "")
    a = agg.setdefault(short, [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
print("--- window totals")
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:40s} {n:6d} launches {us:10.1f} us  avg {us / n:7.2f}")
