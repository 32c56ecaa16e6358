#!/usr/bin/env python3
"""C4 shard on H pooled handles, each driven by its OWN host thread (F batches in flight per handle): what two fully independent
tick streams are worth against one handle's two joined list slices.   usage: lm_threads.py H F [reps per handle]"""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
reg = importlib.import_module("riv-slam_amd.registration"); scene = importlib.import_module("riv-slam_amd.scene")
H, F = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 96
P, N = 32, 8192
LM = dict(max_correspondence_distance=2.0, transformation_epsilon=0.1, azimuth_variance_deg=1.0)
clouds, guesses = [], []
for p in range(P):
    s, t, _, _ = scene.make_pair(N, N, scene.pair_seed(4, p), "loop")
    clouds += [torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()]
    guesses.append(np.eye(4, dtype=np.float32))
torch.cuda.synchronize()
hs = [reg.BatchAPDGICP(reg.default_params(**LM)) for _ in range(H)]
packed = hs[0].pack_clouds(clouds)
pairs = [hs[0].make_pairs([(2 * P * f + 2 * i, 2 * P * f + 2 * i + 1) for i in range(P)], guesses) for f in range(F)]
def run(h, count):
    tk = [None] * F
    for s in range(count):
        f = s % F
        if tk[f] is not None: hs[h].align_collect(tk[f])
        hs[h].set_clouds(2 * P * f, packed, producer_wait=False)
        tk[f] = hs[h].align_enqueue(pairs[f])
    for f in range(F):
        if tk[f] is not None: hs[h].align_collect(tk[f])
    hs[h].synchronize()
def all_(count):
    th = [threading.Thread(target=run, args=(h, count)) for h in range(H)]
    [t.start() for t in th]; [t.join() for t in th]
all_(2 * F)
t0 = time.perf_counter(); all_(reps); dt = time.perf_counter() - t0
print("threads %d x %d in flight: %.3f ms per batch" % (H, F, dt / (reps * H) * 1e3))
