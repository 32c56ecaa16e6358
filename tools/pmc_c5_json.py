#!/usr/bin/env python3
"""PMC numbers of the dense one-pair search launches of C5 (tools/c5_run.py under rocprofv3 --pmc, tools/c5_profile.sh) -> profiles/pmc_c5.json,
stamped with the source stamp compiled into the profiled library like pmc_nn_latest.json (bench.py prints c5_dense.valu_busy only from a file
whose stamp equals the loaded library's).  Averages over the WARM ticks (the launches that are not the first of a registration).
usage: pmc_c5_json.py out.json insts.db busy.db [fetch.db write.db]"""
import importlib, json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
stamp = importlib.import_module("riv-slam_amd.registration").source_stamp()
acc, dur = {}, []
for dbfile in sys.argv[2:]:
    db = sqlite3.connect(dbfile)
    rows = db.execute("select dispatch_id, kernel_name, grid_size_x, counter_name, value, end - start from counters_collection where kernel_name like '%k_nn_pruned%' order by dispatch_id").fetchall()
    by = {}
    for d, k, gx, c, v, t in rows:
        by.setdefault(d, [k, gx, t, {}])[3][c] = v
    ds = sorted(by)
    for i, d in enumerate(ds):
        k, gx, t, cs = by[d]
        if i % 20 == 0:
            continue   # the cold first tick of every GN-20 registration
        for c, v in cs.items():
            a = acc.setdefault(c, [0.0, 0])
            a[0] += v
            a[1] += 1
        dur.append(t / 1e3)
    kernel, grid = (by[ds[-1]][0].split("(")[0].replace("void apd::", ""), by[ds[-1]][1]) if ds else ("", 0)
out = {"source_stamp": stamp, "kernel": kernel, "grid_x": grid, "workload": "tools/c5_run.py: 100k x 500k, GN-20, warm ticks (2 .. 20) of every registration",
       "avg_us_serialised_by_pmc": sum(dur) / max(1, len(dur)), "source": "profiles/pmc_c5.json (tools/pmc_c5_json.py over the rocprofv3 --pmc passes of tools/c5_profile.sh)"}
out.update({c: a[0] / a[1] for c, a in acc.items()})
if "SQ_ACTIVE_INST_VALU" in out:
    out["valu_busy_of_the_launch_alone"] = out["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * 2.4e9 * out["avg_us_serialised_by_pmc"] * 1e-6)
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out))
