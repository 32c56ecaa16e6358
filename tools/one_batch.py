#!/usr/bin/env python3
"""One GN-20 batch of the bench's pairs on one handle (for profilers).  usage: one_batch.py [pairs] [kind] [reps]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
reg = importlib.import_module("riv-slam_amd.registration")
scene = importlib.import_module("riv-slam_amd.scene")
import bench
P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
kind = sys.argv[2] if len(sys.argv) > 2 else "odometry"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
clouds, pairs, guesses = [], [], []
for p in range(P):
    s, t, _, g = scene.make_pair(8192, 8192, scene.pair_seed(2, p), kind)
    clouds += [torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()]
    pairs.append((2 * p, 2 * p + 1))
    guesses.append(g)
b = reg.BatchAPDGICP(bench.bench_params(reg))
b.set_pair_groups(1)
for _ in range(reps):
    b.set_clouds(0, clouds)
    if os.environ.get("APDGICP_STATS"):
        b.compute_covariances()
        b.debug_stats()   # (reading resets: the counters below are the ticks' alone)
    r = b.align(pairs, guesses)
print("done", int(r["n_linearize"].min()))
if os.environ.get("APDGICP_STATS"):
    st = b.debug_stats().astype(float)
    w = max(st[3], 1.0)
    print("stats per wave (all ticks of the last align%s): groups %.2f chunk tests %.2f chunk scans %.2f batches %.2f kept %.1f%% waves %d" %
          (" + covariances" if reps == 1 else "s", st[0] / w, st[1] / w, st[2] / w, st[5] / w, 100 * st[6] / (w * 64), int(w)))
    if st[14] > 0:   # sampled search waves (nn_search): cycles up to the wave box / up to the candidate groups / scans + index + record
        print("search wave, cycles: warm start %.0f  group boxes %.0f  scans, index, record %.0f  (%d sampled waves)" %
              (st[10] / st[14], st[11] / st[14], st[12] / st[14], int(st[14])))
