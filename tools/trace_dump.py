#!/usr/bin/env python3
"""Prints the last N kernels of a rocprofv3 --kernel-trace rocpd database with start offsets, durations and the gap to
the previous kernel's end.  usage: trace_dump.py results.db [N]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = db.execute("select name, start, end, grid_x, grid_y from kernels order by start").fetchall()[-n:]
t0, prev = rows[0][1], rows[0][1]
for name, a, b, gx, gy in rows:
    print("%9.1f us  +%6.1f gap  %7.1f us  %-40s grid (%d,%d)" % ((a - t0) / 1e3, (a - prev) / 1e3, (b - a) / 1e3, name.split("(")[0].replace("void apd::", "")[:40], gx, gy))
    prev = b
