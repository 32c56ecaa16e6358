#!/usr/bin/env python3
"""Latency of the small configurations (C3: one fresh scan against 8 cached keyframes, LM launch parameters / GN-20; C2: one pair, LM launch, target
cached) from the tree given as argv[1] -- so that two checkouts (e.g. a `git worktree` of the previous round inside the repository) can be alternated
on ONE box: python tools/c3_latency.py . ; python tools/c3_latency.py _r05wt"""
import importlib
import os
import sys
import time

import numpy as np

root = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else ".")
sys.path.insert(0, root)
import torch  # noqa: E402

reg = importlib.import_module("riv-slam_amd.registration")
scene = importlib.import_module("riv-slam_amd.scene")
LM_LAUNCH = dict(max_correspondence_distance=2.0, transformation_epsilon=0.1, azimuth_variance_deg=1.0)   # the launch file's registration parameters
GN = dict(optimizer=1, max_iterations=20, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0, azimuth_variance_deg=1.0)


def med(fn, n=300):
    for _ in range(20):
        fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return float(np.median(t)) * 1e3, float(np.percentile(t, 10)) * 1e3


src3, tgts3, _, gs3 = scene.make_keyframe_set(8192, 8192, 8, scene.pair_seed(3, 0))
d3 = torch.from_numpy(src3).cuda()
row = [os.path.basename(root) or "."]
for tag, kw in (("C3 lm_launch", LM_LAUNCH), ("C3 gn20", GN)):
    b = reg.BatchAPDGICP(reg.default_params(**kw))
    src_i = b.add_cloud(d3)
    tg = [b.add_cloud(torch.from_numpy(t).cuda()) for t in tgts3]
    b.compute_covariances()
    pairs = b.make_pairs([(src_i, k) for k in tg], gs3)

    def c3():
        b.set_cloud(src_i, d3)
        return b.align(pairs)
    m, p10 = med(c3)
    row.append("%s %.4f (p10 %.4f)" % (tag, m, p10))
    del b
s, t, _, g = scene.make_pair(8192, 8192, scene.pair_seed(2, 0), "odometry")
ds, dt = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
h = reg.FastAPDGICP(reg.default_params(**LM_LAUNCH))
h.setInputTarget(dt)


def c2():
    h.setInputSource(ds)
    return h.align(g)
m, p10 = med(c2)
row.append("C2 lm_launch target cached %.4f (p10 %.4f)" % (m, p10))
print("  ".join(row))
