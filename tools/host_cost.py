#!/usr/bin/env python3
"""Host time of one bench step's submission (set_clouds + align_enqueue, Gauss-Newton-20: ~90 launches) against the step's
wall time with four handles in flight.  usage: host_cost.py [steps] [--host-clouds]
--host-clouds: every step hands over 64 HOST clouds (numpy, pageable) like `bench.py --host-clouds`; the submit time is split into
set_clouds (packing into pinned memory on the host pool + the asynchronous copy) and align_enqueue (~90 launches)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
reg = importlib.import_module("riv-slam_amd.registration"); scene = importlib.import_module("riv-slam_amd.scene")
import bench
P, H = 32, 4
host = "--host-clouds" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = int(argv[0]) if argv else 200
clouds, pairs, guesses = [], [], []
for p in range(P):
    s, t, _, g = scene.make_pair(8192, 8192, scene.pair_seed(2, p), "odometry")
    clouds += [s, t] if host else [torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()]
    pairs.append((2 * p, 2 * p + 1)); guesses.append(g)
hs = [reg.BatchAPDGICP(bench.bench_params(reg)) for _ in range(H)]
for b in hs: b.set_pair_groups(1)
arg = hs[0].pack_clouds(clouds); arr = hs[0].make_pairs(pairs, guesses)
def run(n):
    tk = [None] * H; t_sub = t_col = t_set = 0.0
    for s in range(n):
        h = s % H
        if tk[h] is not None:
            t0 = time.perf_counter(); hs[h].align_collect(tk[h], device=True); t_col += time.perf_counter() - t0
        t0 = time.perf_counter()
        hs[h].set_clouds(0, arg, producer_wait=False); t1 = time.perf_counter(); tk[h] = hs[h].align_enqueue(arr)
        t_sub += time.perf_counter() - t0; t_set += t1 - t0
    for h in range(H):
        if tk[h] is not None: hs[h].align_collect(tk[h], device=True)
    torch.cuda.synchronize()
    return t_sub, t_col, t_set
run(20)
t0 = time.perf_counter(); t_sub, t_col, t_set = run(steps); wall = time.perf_counter() - t0
print("%s clouds, steps %d: wall %.3f ms per step; host submit (set_clouds + enqueue) %.3f ms per step (set_clouds %.3f, enqueue %.3f), host waiting in collect %.3f ms per step" %
      ("HOST" if host else "resident", steps, wall / steps * 1e3, t_sub / steps * 1e3, t_set / steps * 1e3, (t_sub - t_set) / steps * 1e3, t_col / steps * 1e3))
