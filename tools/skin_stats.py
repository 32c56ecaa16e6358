#!/usr/bin/env python3
"""Per-iteration effect of neighbour keeping (nn_search): runs the bench's GN workload with max_iterations = 1..K and
differences the cumulative pruning counters.   usage (inside gpurun): python tools/skin_stats.py [kind] [pairs]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["APDGICP_STATS"] = "1"
import torch  # noqa
reg = importlib.import_module("riv-slam_amd.registration")
scene = importlib.import_module("riv-slam_amd.scene")
kind = sys.argv[1] if len(sys.argv) > 1 else "odometry"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
clouds, pairs, guesses = [], [], []
for p in range(P):
    s, t, _, g = scene.make_pair(8192, 8192, scene.pair_seed(2, p), kind)
    clouds += [torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()]
    pairs.append((2 * p, 2 * p + 1))
    guesses.append(g)
prev = np.zeros(16)
print("iter  kept%   groups/wave  chunks tested/wave  chunks scanned/wave")
for k in list(range(1, 9)) + [10, 12, 16, 20]:
    prm = reg.default_params(optimizer=reg.OPT_GN, max_iterations=k, transformation_epsilon=1e-300, rotation_epsilon=1e-300,
                             max_correspondence_distance=2.0, azimuth_variance_deg=1.0)
    b = reg.BatchAPDGICP(prm)
    b.set_pair_groups(1)
    b.set_clouds(0, clouds)
    b.compute_covariances()
    b.debug_stats()
    b.align(pairs, guesses)
    st = b.debug_stats().astype(np.float64)
    d = st - prev
    waves = max(d[3], 1.0)
    print(f"{k:3d}  {100 * d[6] / (waves * 64):6.2f}  {d[0] / waves:8.2f}  {d[1] / waves:10.2f}  {d[2] / waves:10.2f}   (waves {int(d[3])})")
    prev = st
