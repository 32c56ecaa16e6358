"""Scan-to-scan odometry as the nodelet runs it (launch parameters: LM, transformation_epsilon 0.1): a new 8192-point scan
against the cached previous keyframe, one registration at a time.  For rocprofv3 --kernel-trace.
usage: python tools/c2_lm_run.py [reps]"""
import importlib, sys, time
sys.path.insert(0, ".")
import torch
reg = importlib.import_module("riv-slam_amd.registration"); scene = importlib.import_module("riv-slam_amd.scene")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
s, t, _, g = scene.make_pair(8192, 8192, scene.pair_seed(2, 0), "odometry")
ds, dt = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
h = reg.FastAPDGICP(reg.default_params(max_correspondence_distance=2.0, transformation_epsilon=0.1, azimuth_variance_deg=1.0))
h.setInputTarget(dt, token=1)
def step(i):
    h.setInputSource(ds, token=100 + i)   # a new scan every call
    return h.align(g)
for i in range(5): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(reps): step(5 + i)
torch.cuda.synchronize()
print("LM (launch parameters), target cached: %.3f ms per registration, n_linearize %d" % ((time.perf_counter() - t0) / reps * 1e3, h.result.n_linearize))
