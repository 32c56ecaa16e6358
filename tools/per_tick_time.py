#!/usr/bin/env python3
"""Duration of every batch search / per-point launch of a profiled GN-20 run in launch order (rocprofv3 --kernel-trace, no counters: the
launches of different handles overlap as in the bench; with --pmc they are serialised: the kernel alone).  usage: per_tick_time.py results.db [first] [count]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = db.execute("select name, grid_y, start, end from kernels where (name like '%k_nn_compact%' or name like '%k_linearize%') and grid_y >= 8 order by start").fetchall()
nn = [(e - s) / 1e3 for n, gy, s, e in rows if "k_nn_compact" in n][first:first + count]
li = [(e - s) / 1e3 for n, gy, s, e in rows if "k_linearize" in n][first:first + count]
print("search us:   ", " ".join(f"{v:.1f}" for v in nn), " sum %.1f" % sum(nn))
print("linearize us:", " ".join(f"{v:.1f}" for v in li), " sum %.1f" % sum(li))
