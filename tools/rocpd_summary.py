#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd SQLite result into the per-kernel tables committed under profiles/.
One row per (kernel, grid): the batch launches of the timed region and the single-pair launches of the other legs of
bench.py are different grids of the same kernels and must not be averaged together.
usage: rocpd_summary.py results.db [title] > profiles/xxx.md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]


def short(name):
    return name.split("(")[0].replace("void ", "")


rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                  "max(vgpr_count), max(sgpr_count), max(lds_size), grid_x, grid_y, grid_z, max(workgroup_x) "
                  "from kernels group by name, grid_x, grid_y, grid_z order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
print(f"# {title}\n")
print("rocprofv3 kernel trace; durations in microseconds; one row per kernel and grid size (work-items); vgpr / sgpr as rocprofv3 reports them "
      "(its vgpr figure is half the compiler's allocation: k_nn_compact<4> 48 here = 96 registers, occupancy 5 -- tools/isa_stats.py has the compiler's)\n")
print("| kernel | calls | total us | avg us | min us | max us | % | vgpr | sgpr | lds B | grid (x,y,z) | wg |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {short(r[0])} | {r[1]} | {r[2]/1e3:.1f} | {r[3]/1e3:.2f} | {r[4]/1e3:.2f} | {r[5]/1e3:.2f} | {100*r[2]/total:.1f} | {r[6]} | {r[7]} | {r[8]} | "
          f"({r[9]},{r[10]},{r[11]}) | {r[12]} |")
try:
    pm = db.execute("select kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name, avg(value), count(*), avg(end-start) "
                    "from counters_collection group by kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name "
                    "order by kernel_name, grid_size_y desc, counter_name").fetchall()
except Exception:  # no counters in this run
    pm = []
if pm:
    counters = sorted({r[4] for r in pm})
    table = {}
    for r in pm:
        table.setdefault((short(r[0]), r[1], r[2], r[3]), {})[r[4]] = (r[5], r[6], r[7])
    print("\n## PMC counters (average per dispatch; one row per kernel and grid)\n")
    print("| kernel | grid (x,y,z) | dispatches | avg us (serialised by --pmc) | " + " | ".join(counters) + " |")
    print("|---|---|---|---|" + "---|" * len(counters))
    for (k, gx, gy, gz), vals in sorted(table.items(), key=lambda kv: -max(v[0] for v in kv[1].values())):
        any_v = next(iter(vals.values()))
        cells = " | ".join(f"{vals[c][0]:.1f}" if c in vals else "" for c in counters)
        print(f"| {k} | ({gx},{gy},{gz}) | {any_v[1]} | {any_v[2]/1e3:.2f} | {cells} |")
