"""Times scan-to-submap target assembly (f3): K keyframe clouds of N points -> transform, concatenate, VoxelGrid.
usage: python tests/measure/bench_submap.py [K] [N] [leaf];  prints device ms (inputs resident in HBM) and the CPU oracle's ms"""
import importlib, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np, torch
sub = importlib.import_module("riv-slam_amd.submap"); scene = importlib.import_module("riv-slam_amd.scene")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
leaf = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
rng = np.random.default_rng(0)
clouds, odoms, T = [], [], np.eye(4)
for f in range(K + 1):
    s, _, Tt, _ = scene.make_pair(N, 16, scene.pair_seed(77, f), "odometry")
    clouds.append(np.ascontiguousarray(np.concatenate([s[:, :3], rng.uniform(0, 40, (N, 1)).astype(np.float32)], 1)))
    T = T @ Tt; odoms.append(T.copy())
poses = sub.relative_poses(odoms[:-1], odoms[-1])
dev = [torch.from_numpy(c).cuda() for c in clouds[:-1]]
a = sub.SubmapAssembler()
ts = []
for it in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = a.assemble(dev, poses, leaf)
    ts.append((time.perf_counter() - t0) * 1e3)
print("K=%d N=%d leaf=%.2f -> %d points; device ms (min of 10, incl. the count read-back): %.3f" % (K, N, leaf, n, min(ts)))
import ref as R
t0 = time.perf_counter(); o, _, _ = R.submap_assemble(clouds[:-1], poses, leaf); t1 = time.perf_counter()
print("CPU oracle (1 thread): %.3f ms, %d points" % ((t1 - t0) * 1e3, o.shape[0]))
