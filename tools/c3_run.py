"""BASELINE configs[2] on the device only: one 8192-point scan against 8 cached keyframes, a new scan every call.
usage: python tools/c3_run.py [gn20|lm_launch] [n_keyframes]"""
import importlib, sys, time
sys.path.insert(0, ".")
import torch
reg = importlib.import_module("riv-slam_amd.registration"); scene = importlib.import_module("riv-slam_amd.scene")
mode = sys.argv[1] if len(sys.argv) > 1 else "gn20"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
kw = (dict(optimizer=1, max_iterations=20, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0, azimuth_variance_deg=1.0)
      if mode == "gn20" else dict(max_correspondence_distance=2.0, transformation_epsilon=0.1, azimuth_variance_deg=1.0))
src, tgts, _, gs = scene.make_keyframe_set(8192, 8192, K, scene.pair_seed(3, 0))
d3 = torch.from_numpy(src).cuda()
b = reg.BatchAPDGICP(reg.default_params(**kw))
si = b.add_cloud(d3)
tg = [b.add_cloud(torch.from_numpy(t).cuda()) for t in tgts]
b.compute_covariances()
pairs = b.make_pairs([(si, k) for k in tg], gs)
def step():
    b.set_cloud(si, d3)
    return b.align(pairs)
for _ in range(5): step()
best = 1e9
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
print("%s 1 x %d: %.3f ms per batch" % (mode, K, best))
