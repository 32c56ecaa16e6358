#!/usr/bin/env python3
"""GPU occupancy of the bench steps from a rocprofv3 --kernel-trace rocpd database: a step starts at its
k_pack_points_multi launch; per step: wall time, time with >= 1 / >= 2 / >= 3 kernels running, idle time, and the
gap between the last kernel of the previous step and the first of this one.
usage: timeline.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_y from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "k_pack_points_multi" in r[0]]
print("steps found:", len(starts))
for si in range(len(starts) - 1):
    seg = rows[starts[si]:starts[si + 1]]
    t0, t1 = seg[0][1], rows[starts[si + 1]][1]
    ev = []
    for _, a, b, _ in seg:
        ev.append((a, 1)), ev.append((min(b, t1), -1))
    ev.sort()
    depth, last, busy = 0, t0, [0, 0, 0, 0]
    for t, d in ev:
        busy[min(depth, 3)] += t - last
        depth, last = depth + d, t
    busy[0] += t1 - last
    prev_end = max(r[2] for r in rows[starts[si - 1]:starts[si]]) if si > 0 else t0
    by = {}
    for name, a, b, gy in seg:
        k = name.split("(")[0].replace("void apd::", "")
        by[k] = by.get(k, 0) + (b - a)
    top = sorted(by.items(), key=lambda kv: -kv[1])[:4]
    print("step %d: wall %.3f ms  idle %.3f  1 kernel %.3f  2 %.3f  >=3 %.3f   gap before %.1f us   %s" % (
        si, (t1 - t0) / 1e6, busy[0] / 1e6, busy[1] / 1e6, busy[2] / 1e6, busy[3] / 1e6, (t0 - prev_end) / 1e3,
        "  ".join("%s %.2f" % (k, v / 1e6) for k, v in top)))

# whole-trace occupancy between the first and the last step boundary (steps overlap when several handles are in flight)
if len(starts) >= 4:
    t0, t1 = rows[starts[len(starts) // 2]][1], rows[starts[-1]][1]
    ev = []
    for name, a, b, _ in rows:
        if b <= t0 or a >= t1:
            continue
        ev.append((max(a, t0), 1)), ev.append((min(b, t1), -1))
    ev.sort()
    depth, last, busy = 0, t0, [0] * 8
    for t, d in ev:
        busy[min(depth, 7)] += t - last
        depth, last = depth + d, t
    busy[0] += t1 - last
    nsteps = len(starts) - 1 - len(starts) // 2
    print("second half of the trace: %.3f ms for %d steps = %.3f ms per step; kernels running at once: %s" % (
        (t1 - t0) / 1e6, nsteps, (t1 - t0) / 1e6 / nsteps, "  ".join("%d: %.0f%%" % (k, 100.0 * v / (t1 - t0)) for k, v in enumerate(busy) if v)))
