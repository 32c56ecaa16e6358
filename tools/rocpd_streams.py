#!/usr/bin/env python3
"""Steady-state stream occupancy of a rocprofv3 kernel trace: per HIP stream the share of time a kernel of that stream was
running, its kernels, and the share of time at least one / at least two kernels were running.
The window is anchored on the BATCHES, not on the trace's clock: the per-batch covariance launches (k_knn_cov_coop with grid y >= 8,
one per batch of a pooled LM run) are counted, and the window runs from the start of launch number lo x count to the start of launch
number hi x count (default 0.45 .. 0.90: behind the warm-up and the pool's ramp-up, in front of the drain; round 5's version took the
middle half of the trace's TIME, which begins with the first kernel of the process and covered the ramp-up).  The window's length /
batches inside it is printed as ms per batch: it must agree with the figure the traced program printed itself.
usage: rocpd_streams.py results.db [lo=0.45] [hi=0.90]   (a trace without such launches: the central part of the time axis)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.45
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.90
rows = db.execute("select name, start, end, stream_id, grid_y from kernels order by start").fetchall()
t0, t1 = rows[0][1], max(r[2] for r in rows)
anchors = [r[1] for r in rows if "k_knn_cov_coop" in r[0] and r[4] >= 8]
if len(anchors) >= 20:
    i0, i1 = int(lo * len(anchors)), min(len(anchors) - 1, int(hi * len(anchors)))
    a, b = anchors[i0], anchors[i1]
    print(f"window: batches {i0} .. {i1} of {len(anchors)} (per-batch covariance launches): {(b - a) / 1e6 / (i1 - i0):.4f} ms per batch inside it")
else:
    a, b = t0 + (t1 - t0) * lo, t0 + (t1 - t0) * hi
# a launch counts with the part of it that lies inside the window
rows = [(n, max(s, a), min(e, b), st, gy) for n, s, e, st, gy in rows if e > a and s < b]
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
