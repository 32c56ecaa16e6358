#!/usr/bin/env python3
"""Steady-state stream occupancy of a rocprofv3 kernel trace: for the middle part of the trace, per HIP stream the share of
time a kernel of that stream was running, its kernels, and the share of time at least one / at least two kernels were running.
usage: rocpd_streams.py results.db [keep=0.5]   (keep: the central fraction of the trace that is analysed)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
keep = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = db.execute("select name, start, end, stream_id, grid_y from kernels order by start").fetchall()
t0, t1 = rows[0][1], max(r[2] for r in rows)
a, b = t0 + (t1 - t0) * (0.5 - keep / 2), t0 + (t1 - t0) * (0.5 + keep / 2)
rows = [r for r in rows if r[1] >= a and r[2] <= b]
span = b - a


def short(n):
    return n.split("(")[0].replace("void ", "").replace("apd::", "")


streams = {}
for n, s, e, st, gy in rows:
    streams.setdefault(st, []).append((short(n), s, e, gy))
print(f"window {span / 1e6:.2f} ms of {(t1 - t0) / 1e6:.2f} ms, {len(rows)} launches on {len(streams)} streams")
for st, ks in sorted(streams.items(), key=lambda kv: -sum(e - s for _, s, e, _ in kv[1])):
    busy = sum(e - s for _, s, e, _ in ks)
    per = {}
    for n, s, e, gy in ks:
        p = per.setdefault(n, [0, 0.0, 0])
        p[0] += 1
        p[1] += (e - s) / 1e3
        p[2] += gy
    desc = ", ".join(f"{n} x{c} avg {us / c:.1f} us (grid.y avg {gy / c:.0f})" for n, (c, us, gy) in sorted(per.items(), key=lambda kv: -kv[1][1])[:5])
    print(f"  stream {st}: busy {100 * busy / span:.0f} %  ({len(ks)} launches)  {desc}")
ev = sorted([(s, 1) for _, s, e, _, _ in rows] + [(e, -1) for _, s, e, _, _ in rows])
depth, last, cover = 0, a, [0.0] * 8
for t, d in ev:
    cover[min(depth, 7)] += t - last
    last, depth = t, depth + d
cover[min(depth, 7)] += b - last
print("kernels running at once: " + ", ".join(f"{i}: {100 * c / span:.0f} %" for i, c in enumerate(cover) if c > 0))
