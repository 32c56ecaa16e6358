"""BASELINE configs[5] (100k-point scan against a 500k-point map) on the device only: first align (sort + covariances of
both clouds + 20 GN iterations), then repeated aligns with everything cached.  For rocprofv3 --kernel-trace --stats.
usage: python tools/c5_run.py [n_src] [n_tgt]"""
import importlib, sys, time
sys.path.insert(0, ".")
import torch
reg = importlib.import_module("riv-slam_amd.registration"); scene = importlib.import_module("riv-slam_amd.scene")
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
GN = dict(optimizer=1, max_iterations=20, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0, azimuth_variance_deg=1.0)
s5, t5, _, g5 = scene.make_pair(ns, nt, scene.pair_seed(5, 0), "odometry")
d5s, d5t = torch.from_numpy(s5).cuda(), torch.from_numpy(t5).cuda()
h = reg.FastAPDGICP(reg.default_params(**GN))
for rep in range(3):
    h.setInputTarget(d5t, token=10 + 2 * rep)
    h.setInputSource(d5s, token=11 + 2 * rep)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h.align(g5)
    first = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    for _ in range(3):
        h.align(g5)
    cached = (time.perf_counter() - t0) / 3 * 1e3
    print("first align (sort + covariances + GN-20) %.2f ms   cached %.2f ms (%.3f ms per GN iteration)" % (first, cached, cached / 20))
