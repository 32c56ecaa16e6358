#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (rocpd SQLite) of a multi-handle run: per stream, the share of the window with a kernel
running; how many kernels run at the same time; average duration per kernel.  (Under the tracer kernels of different
streams hardly overlap: use the durations, not the concurrency, as a statement about the un-traced run.)
usage: rocpd_concurrency.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end,stream_id,grid_x,grid_y from kernels order by start").fetchall()
# steady-state window: middle 50% of the trace
t0, t1 = rows[0][1], max(r[2] for r in rows)
a, b = t0 + (t1 - t0) * 0.55, t0 + (t1 - t0) * 0.95
sel = [r for r in rows if r[1] >= a and r[2] <= b]
print("window ms", (b - a) / 1e6, "kernels", len(sel))
streams = {}
for r in sel: streams.setdefault(r[3], []).append(r)
for sid, rs in sorted(streams.items()):
    busy = sum(r[2] - r[1] for r in rs)
    print(f"stream {sid}: {len(rs)} kernels, busy {busy / (b - a):.3f} of the window")
# concurrency histogram
ev = []
for r in sel: ev.append((r[1], 1)); ev.append((r[2], -1))
ev.sort()
cur, last, hist = 0, a, {}
for t, d in ev:
    hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += d
tot = sum(hist.values())
print("concurrent kernels: " + "  ".join(f"{k}: {v / tot:.3f}" for k, v in sorted(hist.items())))
by = {}
for r in sel:
    k = r[0].split('(')[0].replace('void ', '').replace('apd::', '')
    by.setdefault(k, []).append(r[2] - r[1])
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:28s} n={len(v):5d} avg {sum(v) / len(v) / 1e3:8.2f} us  total {sum(v) / 1e6:8.2f} ms")
