#!/usr/bin/env python3
"""Host time of one bench step's submission (set_clouds + align_enqueue, Gauss-Newton-20: ~90 launches) against the step's
wall time with four handles in flight.  usage: host_cost.py [steps]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
reg = importlib.import_module("riv-slam_amd.registration"); scene = importlib.import_module("riv-slam_amd.scene")
import bench
P, H = 32, 4
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
clouds, pairs, guesses = [], [], []
for p in range(P):
    s, t, _, g = scene.make_pair(8192, 8192, scene.pair_seed(2, p), "odometry")
    clouds += [torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()]
    pairs.append((2 * p, 2 * p + 1)); guesses.append(g)
hs = [reg.BatchAPDGICP(bench.bench_params(reg)) for _ in range(H)]
for b in hs: b.set_pair_groups(1)
arg = hs[0].pack_clouds(clouds); arr = hs[0].make_pairs(pairs, guesses)
def run(n):
    tk = [None] * H; t_sub = t_col = 0.0
    for s in range(n):
        h = s % H
        if tk[h] is not None:
            t0 = time.perf_counter(); hs[h].align_collect(tk[h], device=True); t_col += time.perf_counter() - t0
        t0 = time.perf_counter()
        hs[h].set_clouds(0, arg, producer_wait=False); tk[h] = hs[h].align_enqueue(arr)
        t_sub += time.perf_counter() - t0
    for h in range(H):
        if tk[h] is not None: hs[h].align_collect(tk[h], device=True)
    torch.cuda.synchronize()
    return t_sub, t_col
run(20)
t0 = time.perf_counter(); t_sub, t_col = run(steps); wall = time.perf_counter() - t0
print("steps %d: wall %.3f ms per step; host submit (set_clouds + enqueue) %.3f ms per step, host waiting in collect %.3f ms per step" %
      (steps, wall / steps * 1e3, t_sub / steps * 1e3, t_col / steps * 1e3))
