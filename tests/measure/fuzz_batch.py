#!/usr/bin/env python3
"""Consistency fuzz of the BATCH paths against the single-registration path (product against product: the single path is what
tests/measure/fuzz_parity.py holds against the oracle).  Random batches -- 1 .. 14 pairs, clouds of 41 .. 9 000 points in every mix,
clouds shared between pairs, host / device / padded input, Gauss-Newton and Levenberg-Marquardt (the pair pool, several batches in
flight, collected out of order), every regularisation, both transform orders, plain GICP -- must give, pair by pair, the transform
(bit for bit) and the counts (converged, iterations, n_linearize, n_compute_error) of a fresh FastAPDGICP handle on the same
clouds.  Every failure prints its seed: `python tests/measure/fuzz_batch.py 1 <seed>` replays it.
usage: python tests/measure/fuzz_batch.py [seconds=300] [first_seed=0]  -> one JSON object (commit it under profiles/)"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch  # noqa
reg = importlib.import_module("riv-slam_amd.registration"); scene = importlib.import_module("riv-slam_amd.scene")

BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
SEED0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
SIZES = (41, 63, 64, 65, 127, 129, 255, 256, 257, 512, 1000, 1024, 2047, 2048, 2049, 3000, 4096, 4097, 6000, 8192, 8193, 9000)


def make_batch(seed):
    rng = np.random.default_rng(7_000_003 * seed + 5)
    n_pairs = int(rng.integers(1, 15))
    lm = bool(seed % 2)
    kw = dict(max_correspondence_distance=float(rng.choice((1.0, 2.0, 3.4e38))), azimuth_variance_deg=float(rng.choice((0.5, 1.0))),
              flags=int(rng.choice((0, 0, 2, 1))), regularization=int(rng.integers(0, 5)) if seed % 5 == 0 else 3,
              k_correspondences=int(rng.choice((10, 20, 20, 32))))
    if lm:
        kw.update(transformation_epsilon=float(rng.choice((5e-4, 0.1))))
    else:
        kw.update(optimizer=1, max_iterations=int(rng.integers(1, 9)), transformation_epsilon=1e-300, rotation_epsilon=1e-300)
    clouds, pairs, guesses = [], [], []
    for i in range(n_pairs):
        if clouds and rng.random() < 0.25:          # a pair over clouds already there (one scan against several keyframes, :404-423)
            s_, t_ = int(rng.integers(0, len(clouds))), int(rng.integers(0, len(clouds)))
            g = scene.make_transform(rng.uniform(-0.2, 0.2, 3), *rng.uniform(-0.03, 0.03, 3)).astype(np.float32)
        else:
            n, m = int(rng.choice(SIZES)), int(rng.choice(SIZES))
            s, t, _, g = scene.make_pair(n, m, scene.pair_seed(97, 1000 * seed + i), "loop" if rng.random() < 0.3 else "odometry")
            clouds += [s, t]
            s_, t_ = len(clouds) - 2, len(clouds) - 1
        pairs.append((s_, t_)), guesses.append(g)
    kw["k_correspondences"] = min(kw["k_correspondences"], min(len(c) for c in clouds) - 1)
    form = ("host", "device", "padded")[int(rng.integers(0, 3))]
    return kw, clouds, pairs, guesses, form, lm


def as_input(c, form):
    if form == "device":
        return torch.from_numpy(c).cuda()
    if form == "padded":   # pcl::PointXYZI rows
        buf = np.full((len(c), 8), 3.0, dtype=np.float32)
        buf[:, :3] = c
        return buf
    return c


stats = dict(batches=0, pairs=0, gn_batches=0, lm_batches=0, in_flight_collects=0, mismatches=0)
failures = []
t0 = time.time()
seed = SEED0
singles = {}
while time.time() - t0 < BUDGET:
    kw, clouds, pairs, guesses, form, lm = make_batch(seed)
    want = []
    for (s_, t_), g in zip(pairs, guesses):
        h = reg.FastAPDGICP(reg.default_params(**kw))
        h.setInputSource(clouds[s_]); h.setInputTarget(clouds[t_])
        T = h.align(g)
        r = h.result
        want.append((T.copy(), [int(r.converged), int(r.iterations), int(r.n_linearize), int(r.n_compute_error)]))
    b = reg.BatchAPDGICP(reg.default_params(**kw))
    inputs = [as_input(c, form) for c in clouds]
    bad = []
    if seed % 3 == 0:     # several copies of the batch in flight (own cloud slots each), collected newest first
        copies = 3 if lm else 2   # (a Gauss-Newton handle keeps the records of its last two batches)
        tickets = []
        for q in range(copies):
            base = q * len(clouds)
            b.set_clouds(base, inputs)
            tickets.append(b.align_enqueue(b.make_pairs([(base + a, base + c) for a, c in pairs], guesses)))
        results = [b.align_collect(t) for t in reversed(tickets)]
        stats["in_flight_collects"] += copies
    else:
        b.set_clouds(0, inputs)
        results = [b.align(pairs, guesses)]
    for which, res in enumerate(results):
        for i, (T, info) in enumerate(want):
            got = [int(res[i]["converged"]), int(res[i]["iterations"]), int(res[i]["n_linearize"]), int(res[i]["n_compute_error"])]
            if not np.array_equal(reg.result_matrix(res[i]), T) or got != info:
                bad.append(f"copy {which} pair {i} {pairs[i]} sizes {len(clouds[pairs[i][0]])}x{len(clouds[pairs[i][1]])}: counts {got} vs {info}, "
                           f"max |dT| {float(np.abs(reg.result_matrix(res[i]) - T).max()):.3g}")
    stats["batches"] += 1; stats["pairs"] += len(pairs); stats["lm_batches" if lm else "gn_batches"] += 1
    if bad:
        stats["mismatches"] += len(bad)
        failures.append(dict(seed=seed, params=kw, form=form, lm=lm, n_pairs=len(pairs), what=bad[:6]))
        print("FAIL", failures[-1], file=sys.stderr, flush=True)
    seed += 1
out = dict(seeds=[SEED0, seed - 1], seconds=round(time.time() - t0, 1), **stats, failures=failures[:30], n_failures=len(failures))
print(json.dumps(out, indent=1))
sys.exit(1 if failures else 0)
