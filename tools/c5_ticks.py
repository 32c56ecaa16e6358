#!/usr/bin/env python3
"""Per-tick search / per-point launch durations of the LAST cached GN-20 registration in a trace of tools/c5_run.py (the last 20 search
launches over the 100k-point source), and with `pmc` the counters of those dispatches.  usage: c5_ticks.py results.db [pmc]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
if len(sys.argv) > 2:
    rows = db.execute("select dispatch_id, kernel_name, counter_name, value, end - start from counters_collection where kernel_name like '%k_nn_%' order by dispatch_id").fetchall()
    by = {}
    for d, k, c, v, dur in rows:
        by.setdefault(d, [k.split("(")[0].replace("void apd::", ""), dur, {}])[2][c] = v
    last = sorted(by)[-20:]
    for i, d in enumerate(last):
        k, dur, cs = by[d]
        w = max(1.0, cs.get("SQ_WAVES", 1.0))
        print(f"tick {i + 1:2d} {k} {dur / 1e3:7.1f} us (serialised)  waves {w:.0f}  VALU/wave {cs.get('SQ_INSTS_VALU', 0) / w:.0f}  SALU/wave {cs.get('SQ_INSTS_SALU', 0) / w:.0f}  "
              f"VMEM/wave {cs.get('SQ_INSTS_VMEM', 0) / w:.0f}  LDS/wave {cs.get('SQ_INSTS_LDS', 0) / w:.0f}  wave-cycles/wave {cs.get('SQ_WAVE_CYCLES', 0) * 4 / w:.0f}")
    sys.exit(0)
rows = db.execute("select name, grid_x, start, end from kernels where (name like '%k_nn_%' or name like '%k_linearize%') order by start").fetchall()
nn = [((e - s) / 1e3, n.split("(")[0].replace("void apd::", ""), gx) for n, gx, s, e in rows if "k_nn_" in n][-20:]
li = [(e - s) / 1e3 for n, gx, s, e in rows if "k_linearize" in n][-20:]
print("search kernel:", nn[0][1], "grid x", nn[0][2])
print("search us:   ", " ".join(f"{v[0]:.1f}" for v in nn), " sum %.1f" % sum(v[0] for v in nn))
print("linearize us:", " ".join(f"{v:.1f}" for v in li), " sum %.1f" % sum(li))
