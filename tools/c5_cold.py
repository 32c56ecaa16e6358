#!/usr/bin/env python3
"""C5 (100k x 500k) registrations of one / two / three GN iterations with cached covariances: what the cold search tick costs under
APDGICP_NN_W = 4 / 8 / 2 (docs/experiments.md, round 6: four waves per 64 points are the best on the cold tick as well)."""
import importlib, sys, time, os
sys.path.insert(0, ".")
import torch
reg = importlib.import_module("riv-slam_amd.registration"); scene = importlib.import_module("riv-slam_amd.scene")
s5, t5, _, g5 = scene.make_pair(100_000, 500_000, scene.pair_seed(5, 0), "odometry")
d5s, d5t = torch.from_numpy(s5).cuda(), torch.from_numpy(t5).cuda()
for iters in (1, 2, 3):
    GN = dict(optimizer=1, max_iterations=iters, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0, azimuth_variance_deg=1.0)
    h = reg.FastAPDGICP(reg.default_params(**GN))
    h.setInputTarget(d5t, token=10); h.setInputSource(d5s, token=11)
    h.align(g5); h.align(g5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): h.align(g5)
    print("NN_W=%s GN-%d cached: %.3f ms" % (os.environ.get("APDGICP_NN_W", "auto"), iters, (time.perf_counter() - t0) / 10 * 1e3))
