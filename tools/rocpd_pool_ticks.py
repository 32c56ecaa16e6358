#!/usr/bin/env python3
"""Per tick of a pooled-LM kernel trace: pairs covered (grid y) and the durations of search / linearize / error, plus the idle
time of the tick stream.  usage: rocpd_pool_ticks.py results.db [ms=6]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
rows = db.execute("select name, start, end, grid_y from kernels order by start").fetchall()
t_end = rows[-1][2]
rows = [r for r in rows if r[1] >= t_end - ms * 1e6]
tick = [(n.split("(")[0].replace("void ", "").replace("apd::", ""), s, e, y) for n, s, e, y in rows if any(k in n for k in ("k_nn_", "k_linearize", "k_error", "k_pool_poll"))]
busy = sum(e - s for _, s, e, _ in tick)
span = tick[-1][2] - tick[0][1]
print(f"tick stream: {len(tick)} launches, busy {busy / 1e3:.0f} us of {span / 1e3:.0f} us ({100 * busy / span:.0f} %)")
other = {}
for n, s, e, y in rows:
    k = n.split("(")[0].replace("void ", "").replace("apd::", "")
    if not any(x in k for x in ("k_nn_", "k_linearize", "k_error", "k_pool_poll")):
        o = other.setdefault(k, [0, 0.0])
        o[0] += 1
        o[1] += (e - s) / 1e3
print("cloud stream:", ", ".join(f"{k} x{v[0]} {v[1]:.0f} us" for k, v in other.items()))
hist = {}
for n, s, e, y in tick:
    if n.startswith("k_nn_"):
        b = (y // 8) * 8
        h = hist.setdefault((n, b), [0, 0.0])
        h[0] += 1
        h[1] += (e - s) / 1e3
for (n, b), (c, us) in sorted(hist.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"  pairs {b:3d}..{b + 7:3d}  {n:22s} {c:4d} launches  avg {us / c:6.1f} us")
for kname in ("k_linearize<true>", "k_error", "k_pool_poll"):
    v = [(e - s) / 1e3 for n, s, e, _ in tick if n == kname]
    if v:
        print(f"  {kname:22s} {len(v):4d} launches  avg {sum(v) / len(v):6.1f} us")
