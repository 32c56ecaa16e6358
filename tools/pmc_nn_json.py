#!/usr/bin/env python3
"""Per-launch PMC numbers of the dominant kernel (the nearest-neighbour search, batch launches only) from separate
rocprofv3 --pmc passes: writes profiles/pmc_nn_latest.json, which bench.py reports as roofline.traffic / roofline_issue
when the launch shape matches its own.
usage: pmc_nn_json.py out.json points kind pass1.db [pass2.db ...]
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (it reports half of a wide coalesced read)."""
import json
import sqlite3
import sys

out_path, points, kind = sys.argv[1], int(sys.argv[2]), sys.argv[3]
vals, shape = {}, None
for dbfile in sys.argv[4:]:
    db = sqlite3.connect(dbfile)
    # the batch's pair-group launches (grid y = pairs per launch >= 8), not the single-pair legs; one kernel instantiation only
    rows = db.execute("select kernel_name, grid_size_x, grid_size_y, workgroup_size_x, counter_name, avg(value), count(*) from counters_collection "
                      "where kernel_name like '%k_nn_pruned%' and grid_size_y >= 8 group by kernel_name, grid_size_x, grid_size_y, workgroup_size_x, "
                      "counter_name").fetchall()
    shapes = {(r[0].split("(")[0].replace("void ", ""), r[1], r[2], r[3]) for r in rows}
    assert len(shapes) == 1, f"{dbfile}: expected exactly one batch instantiation of k_nn_pruned, found {shapes}"
    assert shape in (None, next(iter(shapes))), (shape, shapes)
    shape = next(iter(shapes))
    for r in rows:
        vals[r[4]] = (float(r[5]), int(r[6]))
kernel, gx, gy, wg = shape
out = {"kernel": kernel, "grid": [gx, gy, 1], "workgroup": wg, "points": points, "pairs_per_launch": gy, "kind": kind, "nn_mode": "pruned",
       "source": "profiles/pmc_nn_latest.json (tools/pmc_nn_json.py over the rocprofv3 --pmc passes of tools/refresh_evidence.sh)",
       "dispatches": {k: v[1] for k, v in vals.items()}}
for k, v in vals.items():
    out[k] = v[0]
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    out["hbm_bytes_per_launch"] = int(round((2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024))
    out["hbm_note"] = "2 x FETCH_SIZE (gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md) + WRITE_SIZE, KB -> bytes"
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out))
