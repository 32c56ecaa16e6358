#!/usr/bin/env python3
"""Where does a bench step go?  The same 32-pair batches on H handles (one pair group each), three workloads:
  full   set 64 fresh clouds + GN-20 (the bench step)
  cov    set 64 fresh clouds + covariances only
  ticks  clouds and covariances cached, GN-20 only
usage (inside gpurun): python tools/phase_bench.py [handles] [pairs] [steps]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
reg = importlib.import_module("riv-slam_amd.registration")
scene = importlib.import_module("riv-slam_amd.scene")
import bench
H = int(sys.argv[1]) if len(sys.argv) > 1 else 4
P = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
clouds, pairs, guesses = [], [], []
for p in range(P):
    s, t, _, g = scene.make_pair(8192, 8192, scene.pair_seed(2, p), "odometry")
    clouds += [torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()]
    pairs.append((2 * p, 2 * p + 1)); guesses.append(g)
prm = bench.bench_params(reg)
hs = []
for _ in range(H):
    b = reg.BatchAPDGICP(prm)
    if H > 1: b.set_pair_groups(1)
    hs.append(b)
arr = hs[0].make_pairs(pairs, guesses)
packed = hs[0].pack_clouds(clouds)
def run(mode, n):
    tick = [None] * H
    for s in range(n):
        b = hs[s % H]
        if mode == "cov":
            b.set_clouds(0, packed); b.compute_covariances()      # (compute_covariances waits for its error flag)
        else:
            if tick[s % H] is not None: b.align_collect(tick[s % H])
            if mode == "full": b.set_clouds(0, packed)
            tick[s % H] = b.align_enqueue(arr)
    for h in range(H):
        if tick[h] is not None: hs[h].align_collect(tick[h])
    torch.cuda.synchronize()
for mode in ("full", "ticks", "cov", "full"):
    if mode == "ticks":
        for b in hs: b.set_clouds(0, packed); b.compute_covariances()
    run(mode, 2 * H); 
    ts = []
    for r in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(mode, steps); ts.append((time.perf_counter() - t0) / steps * 1e3)
    print(f"{mode:6s} handles={H} pairs={P}: {np.median(ts):.4f} ms per step  ({P * 1e3 / np.median(ts):.0f} reg/s)")
