#!/usr/bin/env python3
"""PMC numbers of the pooled-LM tick kernels (bench.py --kind loop --optimizer lm) from separate rocprofv3 --pmc passes: the tick
launches of a pair pool differ in size from launch to launch, so the counters are normalised PER LISTED PAIR SLOT (grid y):
sum over the dispatches of a kernel / sum of their grid y.  Writes profiles/pmc_lm_loop.json (source-stamped like
pmc_nn_latest.json; bench.py refuses a stale file).   usage: pmc_lm_json.py out.json points pass1.db [pass2.db ...]"""
import importlib
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
source_stamp = importlib.import_module("riv-slam_amd.registration").source_stamp()   # compiled into the library that was profiled (apdgicp_source_stamp)
out_path, points = sys.argv[1], int(sys.argv[2])


def short(name):
    return name.split("(")[0].replace("void ", "").replace("apd::", "")


acc = {}   # kernel -> counter -> [sum value, sum grid y, dispatches]
for dbfile in sys.argv[3:]:
    db = sqlite3.connect(dbfile)
    for k, gx, gy, c, v in db.execute("select kernel_name, grid_size_x, grid_size_y, counter_name, value from counters_collection where kernel_name like '%apd::%'"):
        k = short(k)
        if not (k.startswith("k_nn_") or k.startswith("k_linearize") or k == "k_error") or gx != points:
            continue
        a = acc.setdefault(k, {}).setdefault(c, [0.0, 0, 0])
        a[0] += float(v)
        a[1] += int(gy)
        a[2] += 1
nn = [k for k in acc if k.startswith("k_nn_")]
assert nn, "no search launches in the passes"
main = max(nn, key=lambda k: max(a[2] for a in acc[k].values()))
out = {"source_stamp": source_stamp, "kernel": main, "points": points, "kind": "loop", "optimizer": "lm", "nn_mode": "pruned",
       "normalisation": "per listed pair slot: sum over dispatches / sum of grid y (slots behind the end of the list hold -1 and execute nothing)",
       "source": "profiles/pmc_lm_loop.json (tools/pmc_lm_json.py over the rocprofv3 --pmc passes of tools/refresh_evidence.sh)",
       "per_slot": {k: {c: a[0] / max(1, a[1]) for c, a in cs.items()} for k, cs in acc.items()},
       "dispatches": {k: max(a[2] for a in cs.values()) for k, cs in acc.items()},
       "slots_per_dispatch": {k: max(a[1] / a[2] for a in cs.values()) for k, cs in acc.items()}}
ps = out["per_slot"][main]
if "FETCH_SIZE" in ps and "WRITE_SIZE" in ps:
    out["hbm_bytes_per_slot"] = (2.0 * ps["FETCH_SIZE"] + ps["WRITE_SIZE"]) * 1024
    out["hbm_note"] = "2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, KB -> bytes"
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out))
