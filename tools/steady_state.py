#!/usr/bin/env python3
"""The bench step SUSTAINED: H handles, many steps without a synchronisation in between (bench.py's timed region is K = 20
steps between two synchronisations, so its handles start every repetition in phase; here they drift into whatever pattern
the GPU's scheduling gives them), and where the host thread's time goes (collect = waiting for the GPU).
usage (inside gpurun): python tools/steady_state.py [handles=4] [steps=60] [pair groups per handle=1]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
reg = importlib.import_module("riv-slam_amd.registration")
scene = importlib.import_module("riv-slam_amd.scene")
import bench
H = int(sys.argv[1]) if len(sys.argv) > 1 else 4
P = 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
G = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # pair groups (streams) per handle
clouds, pairs, guesses = [], [], []
for p in range(P):
    s, t, _, g = scene.make_pair(8192, 8192, scene.pair_seed(2, p), "odometry")
    clouds += [torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()]
    pairs.append((2 * p, 2 * p + 1)); guesses.append(g)
prm = bench.bench_params(reg)
hs = []
for _ in range(H):
    b = reg.BatchAPDGICP(prm); b.set_pair_groups(G); hs.append(b)
arr = hs[0].make_pairs(pairs, guesses)
packed = hs[0].pack_clouds(clouds)
def run(n, rec=None):
    tick = [None] * H
    for s in range(n):
        b = hs[s % H]
        t0 = time.perf_counter()
        if tick[s % H] is not None: b.align_collect(tick[s % H])
        t1 = time.perf_counter()
        b.set_clouds(0, packed)
        t2 = time.perf_counter()
        tick[s % H] = b.align_enqueue(arr)
        t3 = time.perf_counter()
        if rec is not None: rec.append((t1 - t0, t2 - t1, t3 - t2))
    for h in range(H):
        if tick[h] is not None: hs[h].align_collect(tick[h])
    torch.cuda.synchronize()
run(2 * H)
rec = []
torch.cuda.synchronize(); t0 = time.perf_counter(); run(steps, rec); dt = (time.perf_counter() - t0) / steps * 1e3
a = np.array(rec) * 1e3
print(f"H={H} G={G} steps={steps}: {dt:.4f} ms per step; host per step: collect {a[:,0].mean():.3f} set_clouds {a[:,1].mean():.3f} enqueue {a[:,2].mean():.3f} ms")
for lo in range(0, steps, 10):
    print("  steps %d..: collect %.3f set %.3f enqueue %.3f" % (lo, a[lo:lo+10,0].mean(), a[lo:lo+10,1].mean(), a[lo:lo+10,2].mean()))
