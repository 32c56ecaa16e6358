"""Times ONE optimiser tick from a cold start (no warm-start hints): P pairs, GN, max_iterations = 1, covariances cached.
usage: python tools/cold_tick.py [P] [kind]   (env APDGICP_NN_LANE_SEED=0 for the wave-level seed)"""
import importlib, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
reg = importlib.import_module("riv-slam_amd.registration"); scene = importlib.import_module("riv-slam_amd.scene")
import bench
P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
kind = sys.argv[2] if len(sys.argv) > 2 else "odometry"
prm = bench.bench_params(reg); prm.max_iterations = 1
b = reg.BatchAPDGICP(prm)
g = []
for p in range(P):
    s, t, _, gs = scene.make_pair(8192, 8192, scene.pair_seed(2, p), kind)
    b.add_cloud(s); b.add_cloud(t); g.append(gs)
pairs = [(2 * i, 2 * i + 1) for i in range(P)]
b.align(pairs, g)
ts = []
for it in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter(); b.align(pairs, g); ts.append((time.perf_counter() - t0) * 1e3)
print("P=%d %s: one cold tick (search + linearize + step), ms: min %.3f  all %s" % (P, kind, min(ts), " ".join("%.3f" % x for x in ts)))
