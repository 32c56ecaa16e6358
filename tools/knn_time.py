"""Times calculate_covariances alone (k_knn_cov_*): 64 clouds of 8192 points by default.
usage: python tools/knn_time.py [n_clouds] [n_points]; env APDGICP_STATS=1 prints the per-wave phase timers"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
reg = importlib.import_module("riv-slam_amd.registration"); scene = importlib.import_module("riv-slam_amd.scene")
import bench
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
b = reg.BatchAPDGICP(bench.bench_params(reg))
clouds = []
for p in range(NC // 2):
    s, t, _, _ = scene.make_pair(N, N, scene.pair_seed(2, p), "odometry")
    clouds += [torch.from_numpy(np.ascontiguousarray(s)).cuda(), torch.from_numpy(np.ascontiguousarray(t)).cuda()]
ts = []
for it in range(8):
    b.set_clouds(0, clouds)
    b.synchronize()
    t0 = time.perf_counter()
    b.compute_covariances()
    b.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("sort+knn ms (min of 8): %.3f   all: %s" % (min(ts), " ".join("%.3f" % t for t in ts)))
if os.environ.get("APDGICP_STATS"):
    st = b.debug_stats()
    print("KNN: groups loaded/wave %.1f  (query,group) pairs/wave %.1f  compactions/wave %.2f  waves %d" % (st[4]/st[7], st[9]/st[7], st[8]/st[7], st[7]))
    print("ticks/wave: A %.0f  gneed %.0f  B %.0f  C(rounds) %.0f  C incl. regularisation %.0f" % (st[10]/st[7], st[11]/st[7], st[12]/st[7], st[13]/st[7], st[15]/st[7]))
