#!/usr/bin/env python3
"""Per-launch PMC numbers of the bench's kernels (batch launches only) from separate rocprofv3 --pmc passes: writes
profiles/pmc_nn_latest.json, which bench.py reports as roofline.traffic / roofline_issue when the launch shape matches its own.
usage: pmc_nn_json.py out.json points kind pass1.db [pass2.db ...]
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (it reports half of a wide coalesced read).
The file is stamped with a hash of riv-slam_amd/csrc/* (build.source_stamp): bench.py refuses a file whose stamp is not that of
the sources it runs, so counters of an older kernel can never be divided by the launch times of a newer one."""
import importlib
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
source_stamp = importlib.import_module("riv-slam_amd.registration").source_stamp()   # compiled into the library that was profiled (apdgicp_source_stamp)   # the sources the profiled library was built from

out_path, points, kind = sys.argv[1], int(sys.argv[2]), sys.argv[3]


def short(name):
    return name.split("(")[0].replace("void ", "").replace("apd::", "")


per_kernel = {}   # (kernel, grid) -> {counter: (avg, dispatches)} for the batch launches (grid y = pairs or clouds per launch >= 8)
for dbfile in sys.argv[4:]:
    db = sqlite3.connect(dbfile)
    rows = db.execute("select kernel_name, grid_size_x, grid_size_y, workgroup_size_x, counter_name, avg(value), count(*) from counters_collection "
                      "where kernel_name like '%apd::%' and grid_size_y >= 8 group by kernel_name, grid_size_x, grid_size_y, workgroup_size_x, "
                      "counter_name").fetchall()
    for k, gx, gy, wg, c, v, n in rows:
        per_kernel.setdefault((short(k), gx, gy, wg), {})[c] = (float(v), int(n))
nn = [key for key in per_kernel if key[0].startswith("k_nn_")]
assert len(nn) == 1, f"expected exactly one batch instantiation of the search kernel, found {nn}"
kernel, gx, gy, wg = nn[0]
vals = per_kernel[nn[0]]
out = {"source_stamp": source_stamp, "kernel": kernel, "grid": [gx, gy, 1], "workgroup": wg, "points": points, "pairs_per_launch": gy, "kind": kind, "nn_mode": "pruned",
       "source": "profiles/pmc_nn_latest.json (tools/pmc_nn_json.py over the rocprofv3 --pmc passes of tools/refresh_evidence.sh)",
       "dispatches": {k: v[1] for k, v in vals.items()}}
for k, v in vals.items():
    out[k] = v[0]
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    out["hbm_bytes_per_launch"] = int(round((2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024))
    out["hbm_note"] = "2 x FETCH_SIZE (gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md) + WRITE_SIZE, KB -> bytes"
# every batch kernel of a step: launches per step are the bench's own (20 ticks, one covariance / sort / pack launch)
# (SQ_ACTIVE_INST_VALU: one unit = one quad-cycle = the four cycles a vector instruction holds its SIMD's issue slot -- two and a half
# for plain fp32 add / mul / fma and simple integer operations, tools/ubench_issue.hip; GRBM_GUI_ACTIVE: gfx clocks, summed over the 8 XCDs)
out["step_kernels"] = {f"{k} grid=({gx},{gy})": {c: v[0] for c, v in cs.items() if c.startswith("SQ_INSTS") or c in ("SQ_WAVES", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE")}
                       for (k, gx, gy, wg), cs in sorted(per_kernel.items())}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out))
