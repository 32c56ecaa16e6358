#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd SQLite result (--kernel-trace --stats) into the per-kernel summary table
committed under profiles/.   usage: rocpd_summary.py results.db [title] > profiles/xxx.md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
# one row per (kernel, grid): the batch launches of the timed region and the single-pair launches of the other legs differ
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                  "max(vgpr_count), max(sgpr_count), max(lds_size), grid_x, grid_y, grid_z, max(workgroup_x) "
                  "from kernels group by name, grid_x, grid_y, grid_z order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
print(f"# {title}\n")
print("rocprofv3 --kernel-trace --stats; durations in microseconds; one row per kernel and grid size (work-items)\n")
print("| kernel | calls | total us | avg us | min us | max us | % | vgpr | sgpr | lds B | grid (x,y,z) | wg |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    name = r[0].split("(")[0].replace("void ", "")
    print(f"| {name} | {r[1]} | {r[2]/1e3:.1f} | {r[3]/1e3:.2f} | {r[4]/1e3:.2f} | {r[5]/1e3:.2f} | {100*r[2]/total:.1f} | {r[6]} | {r[7]} | {r[8]} | "
          f"({r[9]},{r[10]},{r[11]}) | {r[12]} |")
try:
    pm = db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name order by kernel_name").fetchall()
    if pm:
        print("\n## PMC counters (average per dispatch)\n\n| kernel | counter | avg | dispatches |\n|---|---|---|---|")
        for r in pm:
            print(f"| {r[0].split('(')[0].replace('void ', '')} | {r[1]} | {r[2]:.1f} | {r[3]} |")
except Exception as e:  # no counters in this run
    pass
