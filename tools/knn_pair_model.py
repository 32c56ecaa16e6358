#!/usr/bin/env python3
"""how many (query, candidate) pairs phase B of the covariance k-NN (k_knn_cov_coop<4>, k = 20) would examine under
smaller groups and a tighter bound -- a numpy model on the bench clouds, BEFORE anything is built.  The kernel's pre-ordering is restated
(apd_sort.hpp: 30-bit Hilbert position in the cloud's bounding cube, ties by index), the shipped phase A too: a wave = 16 consecutive
queries, one 128-point window centred on them, tau = the k-th smallest of the 32 class minima (class of window element e: e mod 32).
  groups   : 128 (shipped) / 64 / 32 consecutive sorted points per box, for the k-NN only
  tau      : 'class' = shipped bound;  'exact' = the exact k-th smallest of the 128 window distances (VERDICT's (ii))
  pairs    : sum over the groups a query needs (fp32 box lower bound <= tau) of the group's size -- what phase B scans
  mask work: box tests per wave as 64-lane trips.  shipped: one trip per query over the 64 group boxes = 16 trips.  amortised (VERDICT's (i)):
             one box-box test per lane of the wave's queries' box inflated by the wave's largest tau against ceil(ngroups / 64) trips of group
             boxes, then one trip per query and 64 SURVIVING groups
usage: python tools/knn_pair_model.py [clouds=4]   -> a table (docs/experiments.md, round 6)"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
scene = importlib.import_module("riv-slam_amd.scene")
K, WIN, QPW = 20, 128, 16


def hilbert30(p):
    lo, hi = p.min(0), p.max(0)
    ext = np.float32(max((hi - lo).max(), 1e-30))
    scale = np.float32(1023.0) / ext
    X = [np.clip((p[:, a] - lo[a]) * scale, 0, 1023).astype(np.uint32) for a in range(3)]
    Q = 512
    while Q > 1:
        P = np.uint32(Q - 1)
        m = (X[0] & Q) != 0
        X[0] = np.where(m, X[0] ^ P, X[0])
        for a in (1, 2):
            m = (X[a] & Q) != 0
            t = (X[0] ^ X[a]) & P
            X0n = np.where(m, X[0] ^ P, X[0] ^ t)
            X[a] = np.where(m, X[a], X[a] ^ t)
            X[0] = X0n
        Q >>= 1
    X[1] ^= X[0]
    X[2] ^= X[1]
    t = np.zeros_like(X[0])
    Q = 512
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ np.uint32(Q - 1), t)
        Q >>= 1
    X = [x ^ t for x in X]

    def expand(v):
        v = v.astype(np.uint64) & 0x3ff
        out = np.zeros_like(v)
        for b in range(10):
            out |= ((v >> b) & 1) << (3 * b)
        return out
    return (expand(X[0]) << 2) | (expand(X[1]) << 1) | expand(X[2])


def lb_point_box(lo, hi, q):  # squared fp32 distance from q to the boxes [lo, hi] (G x 3)
    d = np.maximum(np.maximum(lo - q, q - hi), 0).astype(np.float32)
    return (d * d).sum(1, dtype=np.float32)


def lb_box_box(lo, hi, qlo, qhi):
    d = np.maximum(np.maximum(lo - qhi, qlo - hi), 0).astype(np.float32)
    return (d * d).sum(1, dtype=np.float32)


def model(cloud):
    n = len(cloud)
    key = hilbert30(cloud)
    order = np.lexsort((np.arange(n), key))
    p = cloud[order].astype(np.float32)
    boxes = {}
    for gs in (128, 64, 32, 16):
        ng = (n + gs - 1) // gs
        lo = np.stack([p[g * gs:(g + 1) * gs].min(0) for g in range(ng)])
        hi = np.stack([p[g * gs:(g + 1) * gs].max(0) for g in range(ng)])
        size = np.array([min(gs, n - g * gs) for g in range(ng)])
        boxes[gs] = (lo, hi, size)
    res = {(gs, tk): dict(pairs=0, groups=0, trips_amortised=0, survivors=0, steps=0, big=0, over64=0) for gs in boxes for tk in ("class", "exact")}
    waves = 0
    for base in range(0, n, QPW):
        w0 = min(max(base - (WIN - QPW) // 2, 0), max(n - WIN, 0))
        win = p[w0:w0 + WIN]
        q = p[base:base + QPW]
        d = ((win[None] - q[:, None]) ** 2).astype(np.float32).sum(2, dtype=np.float32)        # [16, 128]
        cls = d.reshape(len(q), WIN // 32, 32).min(1)                                          # class of element e: e mod 32
        tau = {"class": np.sort(cls, 1)[:, K - 1], "exact": np.sort(d, 1)[:, K - 1]}
        waves += 1
        for gs, (lo, hi, size) in boxes.items():
            for tk, tv in tau.items():
                r = res[(gs, tk)]
                surv = lb_box_box(lo, hi, q.min(0), q.max(0)) <= tv.max()
                r["survivors"] += int(surv.sum())
                r["trips_amortised"] += -(-len(lo) // 64) + len(q) * max(1, -(-int(surv.sum()) // 64))
                # hierarchical masks: the 128-point groups that survive the wave's box test, expanded into their sub-groups (lane = sub-group)
                blo, bhi, _ = boxes[128]
                bsurv = int((lb_box_box(blo, bhi, q.min(0), q.max(0)) <= tv.max()).sum())
                r["big"] += bsurv
                r["over64"] += int(bsurv * (128 // gs) > 64)
                for qi in range(len(q)):
                    need = lb_point_box(lo, hi, q[qi]) <= tv[qi]
                    r["pairs"] += int(size[need].sum())
                    r["groups"] += int(need.sum())
                    r["steps"] += -(-int(need.sum()) // (128 // gs))     # query-major steps of 128 candidate slots: 128 / gs sub-groups each
    return n, waves, res


ncl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
acc, tot_q, tot_w = {}, 0, 0
for c in range(ncl):
    s, t, _, _ = scene.make_pair(8192, 8192, scene.pair_seed(2, c // 2), "odometry")
    n, waves, res = model((s, t)[c % 2][:, :3])
    tot_q += n
    tot_w += waves
    for k_, v in res.items():
        a = acc.setdefault(k_, dict(pairs=0, groups=0, trips_amortised=0, survivors=0, steps=0, big=0, over64=0))
        for f in a:
            a[f] += v[f]
print(f"{ncl} bench clouds of 8192 points, k = {K}; shipped = groups of 128, tau from 32 class minima, 16 mask trips per wave")
print("| group size | tau | pairs per query | groups per query | groups surviving the wave's box test | mask trips per wave: per-query (shipped form) | amortised | (query, 128-slot) steps per wave, query-major | 128-groups surviving per wave | waves whose sub-groups exceed 64 lanes |")
print("|---|---|---|---|---|---|---|---|---|---|")
for (gs, tk), a in sorted(acc.items(), key=lambda kv: (-kv[0][0], kv[0][1])):
    ng = -(-8192 // gs)
    print(f"| {gs} | {tk} | {a['pairs'] / tot_q:.0f} | {a['groups'] / tot_q:.2f} | {a['survivors'] / tot_w:.1f} of {ng} | {16 * -(-ng // 64)} | {a['trips_amortised'] / tot_w:.1f} | {a['steps'] / tot_w:.1f} | {a['big'] / tot_w:.1f} | {100.0 * a['over64'] / tot_w:.1f} % |")
