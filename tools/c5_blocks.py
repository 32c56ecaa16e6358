#!/usr/bin/env python3
"""Block timeline of ONE dense search launch of C5 (100k x 500k, APDGICP_STATS=2): per-block durations (distribution), when the blocks start
(rounds) and what an ideal packing of the same blocks onto the same number of slots would take (longest first / in index order).
usage: APDGICP_STATS=2 python tools/c5_blocks.py [tick=10]   with a library built by `python tools/build_variant.py timeline -DAPD_BLOCK_TIMELINE`
(the product kernel does not carry the time stamps: they cost it a wave per SIMD)"""
import importlib, os, sys
sys.path.insert(0, ".")
os.environ["APDGICP_STATS"] = "2"
import heapq
import numpy as np, torch
reg = importlib.import_module("riv-slam_amd.registration"); scene = importlib.import_module("riv-slam_amd.scene")
tick = int(sys.argv[1]) if len(sys.argv) > 1 else 10
GN = dict(optimizer=1, max_iterations=tick, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0, azimuth_variance_deg=1.0)
s5, t5, _, g5 = scene.make_pair(100_000, 500_000, scene.pair_seed(5, 0), "odometry")
b = reg.BatchAPDGICP(reg.default_params(**GN))
b.set_clouds(0, [torch.from_numpy(s5).cuda(), torch.from_numpy(t5).cuda()])
for _ in range(3):
    b.align([(0, 1)], [g5])
tl = b.debug_block_timeline()
tl = tl[tl[:, 1] > 0]
t0 = tl[:, 0].min()
start = (tl[:, 0] - t0).astype(np.float64) / 100.0   # us
end = (tl[:, 1] - t0).astype(np.float64) / 100.0
dur = end - start
print(f"tick {tick}: {len(tl)} blocks, launch span {end.max():.1f} us; block duration us: mean {dur.mean():.1f} median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} p99 {np.percentile(dur, 99):.1f} max {dur.max():.1f}; sum {dur.sum():.0f} us")
first = start < 2.0
print(f"blocks started in the first 2 us: {first.sum()} (the resident set); last block starts at {start.max():.1f} us; blocks starting after 10 us: {(start > 10).sum()}")
order = np.argsort(start)
late = order[-20:]
print("the 20 blocks that END last: " + " ".join(f"[b{int(tl[i, 2] & 0xffffffff)} start {start[i]:.0f} dur {dur[i]:.0f}]" for i in np.argsort(end)[-20:]))
slots = int(first.sum())
def pack(durs):
    h = [0.0] * slots
    heapq.heapify(h)
    for d in durs:
        heapq.heappush(h, heapq.heappop(h) + d)
    return max(h)
idx_order = dur[np.argsort(tl[:, 2] & np.uint64(0xffffffff))]
print(f"list scheduling of these durations on {slots} slots: in index order {pack(idx_order):.1f} us, longest first {pack(np.sort(dur)[::-1]):.1f} us, lower bound sum / slots {dur.sum() / slots:.1f} us (durations measured under contention: a guide, not a prediction)")
# where the heavy blocks are along the curve
bi = (tl[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
heavy = bi[dur > 2 * np.median(dur)]
print(f"blocks slower than 2 x median: {len(heavy)}; by tenth of the curve: {np.histogram(heavy, bins=10, range=(0, bi.max() + 1))[0].tolist()}")
