#!/usr/bin/env python3
"""Timeline of the LAST launches of a rocprofv3 kernel trace (rocpd SQLite): start relative to the first one shown,
duration, and the idle gap in front of each launch -- where a latency-bound frame spends its time.
usage: rocpd_timeline.py results.db [launches=24]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = db.execute("select name, start, end, grid_x, grid_y, workgroup_x, stream_id from kernels order by start desc limit ?", (n,)).fetchall()[::-1]
t0, prev_end = rows[0][1], None
print("| kernel | grid | stream | start us | dur us | gap us |")
print("|---|---|---|---|---|---|")
for name, s, e, gx, gy, wg, sid in rows:
    gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:.2f}"
    print(f"| {name.split('(')[0].replace('void ', '')} | {gx // max(wg, 1)}x{gy} | {sid} | {(s - t0) / 1e3:.2f} | {(e - s) / 1e3:.2f} | {gap} |")
    prev_end = e if prev_end is None else max(prev_end, e)
print(f"\nspan {(max(r[2] for r in rows) - t0) / 1e3:.2f} us, busy {sum(r[2] - r[1] for r in rows) / 1e3:.2f} us")
