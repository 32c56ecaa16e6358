"""Experiment: two independent batch engines driven from two threads (their steps overlap at random phases) against one
engine keeping two steps in flight.  usage: python tools/two_engines.py [steps]   env APDGICP_STREAMS, GPU_MAX_HW_QUEUES"""
import importlib, sys, threading, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
reg = importlib.import_module("riv-slam_amd.registration"); scene = importlib.import_module("riv-slam_amd.scene")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
NE = int(sys.argv[2]) if len(sys.argv) > 2 else 2
P, n = 32, 8192
clouds, guesses = [], []
for p in range(P):
    s, t, _, g = scene.make_pair(n, n, scene.pair_seed(2, p), "odometry")
    clouds += [torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()]; guesses.append(g)
def make():
    b = reg.BatchAPDGICP(bench.bench_params(reg))
    return b, b.make_pairs([(2 * i, 2 * i + 1) for i in range(P)], guesses), b.pack_clouds(clouds)
def run_sync(b, pairs, packed, count):
    for _ in range(count):
        b.set_clouds(0, packed); b.align_device(pairs)
def run_pipe(b, pairs, packed, count):
    prev = None
    for _ in range(count):
        b.set_clouds(0, packed); t = b.align_enqueue(pairs)
        if prev is not None: b.align_collect(prev, device=True)
        prev = t
    b.align_collect(prev, device=True)
engines = [make() for _ in range(NE)]
e1 = engines[0]
for e in engines: run_sync(*e, 5)
torch.cuda.synchronize()
t0 = time.perf_counter(); run_pipe(*e1, steps); torch.cuda.synchronize(); one = (time.perf_counter() - t0) / steps * 1e3
for mode, fn in (("sync", run_sync), ("pipelined", run_pipe)):
    th = [threading.Thread(target=fn, args=(*e, steps)) for e in engines]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; torch.cuda.synchronize()
    two = (time.perf_counter() - t0) / (NE * steps) * 1e3
    print("one engine, two steps in flight: %.3f ms per step;  %d engines (%s each): %.3f ms per step" % (one, NE, mode, two))
