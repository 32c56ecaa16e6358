#!/usr/bin/env python3
"""Generates tests/golden/apdgicp_golden.npz -- known-answer fixtures for the APD-GICP hot path.

The reference ships no fixture for FastAPDGICP ("parity unpinned"), so these
vectors come from the two build-owned restatements under oracle/ (numpy brute-force and C++ kd-tree),
which must agree with each other before a vector is written:
  * discrete outputs (correspondences, fp32 squared distances, iteration counts, converged flags): exactly;
  * covariances: <= 1e-12 absolute; M, H, b, cost: <= 1e-10 relative (both evaluate the three fp32 angles of the sensor model,
    fast_apdgicp_impl.hpp:168,172-173, with the C library's atan2f algorithm -- include/apd_atan2f.h and its numpy twin -- so what
    is left is the order of fp64 sums and the linear solvers);
  * per-iteration traces (lambda and rho of every LM trial, the pose behind every outer iteration): <= 1e-7 relative / 1e-9 m;
  * final transforms: <= 1e-7 m / 1e-7 rad.
The values stored are the C++ restatement's.  Every vector that depends on the fp32 order of `T * p`
(fast_apdgicp_impl.hpp:149; oracle/apdgicp_ref.cpp:xf_row) exists twice: the plain key holds the default order
(pairwise, Eigen >= 3.3), the key suffixed `_xflin` the linear chain of Eigen 3.2 (flags bit 1,
APDGICP_FLAG_XF_LINEAR_CHAIN); inputs (clouds, poses, guesses) are shared.  Run:  python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
scene = importlib.import_module("riv-slam_amd.scene")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import apdgicp_np as O  # noqa: E402
import ref as R  # noqa: E402
from trace_util import trace_close  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "apdgicp_golden.npz")
LAUNCH = dict(max_correspondence_distance=2.0, transformation_epsilon=0.1, azimuth_variance_deg=1.0)  # launch:91-101


XF = (("", 0), ("_xflin", 2))   # key suffix, params.flags
TOL = 1e-10                     # the two restatements on M, H, b, cost (relative)
worst = {}


def check_traces(tag, r, n):
    """The per-iteration traces of the two restatements (C++: ref_get_trace; numpy: Trace) against each other; returns the C++ one."""
    tr = r.trace()
    tn = {"lambda": np.array(n.trace.lambdas), "rho": np.array(n.trace.rhos), "y0": np.array(n.trace.y0s)[:len(n.trace.rhos)] if n.p.optimizer == 0 else np.zeros(0),
          "yi": np.array(n.trace.yis), "poses": np.array(n.trace.poses).reshape(-1, 4, 4)}
    d = trace_close(tr, tn)
    for k, v in d.items():
        worst["trace_" + k] = max(worst.get("trace_" + k, 0), v)
    assert max(d.values()) < 1.0, (tag, d)
    return tr


def np_params(**kw):
    return O.Params(**kw)


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


def main():
    g = {}
    # ---------------------------------------------------------------- (i) KAT-cov, 256 points
    src, tgt, _, _ = scene.make_pair(2048, 2048, scene.pair_seed(1, 0), "odometry")
    c256 = np.ascontiguousarray(src[:256])
    g["cov_cloud"] = c256
    for name, reg in (("none", 0), ("min_eig", 1), ("norm_min_eig", 2), ("plane", 3), ("frobenius", 4)):
        r = R.RefAPDGICP(R.default_params(regularization=reg))
        r.setInputSource(c256)
        a = r.covariances("source")
        b = O.calculate_covariances(c256, 20, reg)
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max()), (name, np.abs(a - b).max())
        g[f"cov_{name}"] = a
    g["cov_knn_idx"] = O.knn(c256, 20)

    # ---------------------------------------------------------------- (ii) KAT-lin, 2048-pt pair, 3 poses
    src, tgt, T_true, guess = scene.make_pair(2048, 2048, scene.pair_seed(1, 1), "odometry")
    g["lin_source"], g["lin_target"], g["lin_T_true"], g["lin_guess"] = src, tgt, T_true, guess
    poses = [np.eye(4), guess.astype(np.float64), T_true]
    for tag, kw0, sfx, fl in [(t, k, s_, f) for t, k in (("default", {}), ("launch", LAUNCH)) for s_, f in XF]:
        kw = dict(kw0, flags=fl)
        tag = tag + sfx
        r = R.RefAPDGICP(R.default_params(**kw))
        n = O.FastAPDGICP(np_params(**kw))
        for o in (r, n):
            o.setInputSource(src)
            o.setInputTarget(tgt)
        n.source_covs = O.calculate_covariances(src)
        n.target_covs = O.calculate_covariances(tgt)
        for k, T in enumerate(poses):
            cr, Hr, br = r.linearize(T)
            cn, Hn, bn = n.linearize(T)
            corr, sqd = r.correspondences()
            assert np.array_equal(corr, n.correspondences) and np.array_equal(sqd, n.sq_distances)
            worst["lin"] = max(worst.get("lin", 0), rel(Hr, Hn), rel(br, bn), abs(cr - cn) / cn)
            assert rel(Hr, Hn) < TOL and rel(br, bn) < TOL and abs(cr - cn) < TOL * cn, (rel(Hr, Hn), rel(br, bn))
            M = r.mahalanobis()
            worst["maha"] = max(worst.get("maha", 0), rel(M, n.mahalanobis))
            assert rel(M, n.mahalanobis) < TOL
            Tt = poses[(k + 1) % 3]
            er, en = r.compute_error(Tt), n.compute_error(Tt)
            assert abs(er - en) < TOL * en
            g[f"lin_{tag}_{k}_T"] = T
            g[f"lin_{tag}_{k}_corr"], g[f"lin_{tag}_{k}_sqd"] = corr, sqd
            g[f"lin_{tag}_{k}_H"], g[f"lin_{tag}_{k}_b"], g[f"lin_{tag}_{k}_cost"] = Hr, br, cr
            g[f"lin_{tag}_{k}_maha128"] = M[:128]
            g[f"lin_{tag}_{k}_errT"], g[f"lin_{tag}_{k}_err"] = Tt, er
    g["lin_source_cov"] = r.covariances("source")
    g["lin_target_cov"] = r.covariances("target")

    # ---------------------------------------------------------------- (iii) KAT-lm: full optimisation
    runs = {
        "lm_default": {},
        "lm_launch": LAUNCH,
        "gn20": dict(optimizer=1, max_iterations=20, transformation_epsilon=1e-300, rotation_epsilon=1e-300),
        "lm_loop": dict(max_correspondence_distance=2.5),   # loop-closure style: identity guess
    }
    for tag, kw0, sfx, fl in [(t, k, s_, f) for t, k in runs.items() for s_, f in XF]:
        kw = dict(kw0, flags=fl)
        if tag == "lm_loop":
            s, t, Tt, gs = scene.make_pair(2048, 2048, scene.pair_seed(1, 2), "loop")
        else:
            s, t, Tt, gs = src, tgt, T_true, guess
        if tag == "lm_loop":  # the other runs reuse lin_source / lin_target / lin_guess
            g[f"{tag}_source"], g[f"{tag}_target"], g[f"{tag}_guess"] = s, t, gs
        tag = tag + sfx
        r = R.RefAPDGICP(R.default_params(**kw))
        n = O.FastAPDGICP(np_params(**kw))
        for o in (r, n):
            o.setInputSource(s)
            o.setInputTarget(t)
        Tr, Tn = r.align(gs), n.align(gs)
        te, re_ = scene.pose_error(Tr, Tn)
        worst["final_pose"] = max(worst.get("final_pose", 0), te, re_)
        assert te < 1e-7 and re_ < 1e-7, (tag, te, re_)
        assert (r.converged, r.nr_iterations, r.n_linearize, r.n_compute_error) == (
            n.converged, n.nr_iterations, n.trace.n_linearize, n.trace.n_compute_error), tag
        g[f"{tag}_T"] = Tr
        g[f"{tag}_info"] = np.array([r.converged, r.nr_iterations, r.n_linearize, r.n_compute_error], dtype=np.int32)
        g[f"{tag}_final_hessian"] = r.final_hessian()
        for k_, v_ in check_traces(tag, r, n).items():
            g[f"{tag}_trace_{k_}"] = v_
        print(tag, "converged", r.converged, "iters", r.nr_iterations, "lin", r.n_linearize, "err", r.n_compute_error,
              "rho<0:", int((np.array(n.trace.rhos) < 0).sum()), "vs truth", scene.pose_error(Tt, Tr))

    # ---------------------------------------------------------------- (iv) degenerate cases
    # (a) a source point that lands on the sensor +x axis (APD 1/cos(AoA) blow-up, :168-171) and
    #     points beyond the correspondence gate
    s = src[:512].copy()
    t = tgt[:512].copy()
    s[0] = (30.0, 0.01, 0.0)
    s[1] = (55.0, 0.0, 1e-3)
    t[0] = (30.05, 0.02, 0.01)
    t[1] = (55.02, 0.01, 0.0)
    s[2] = (10.0, 60.0, 40.0)  # nothing within 2 m: unmatched
    g["deg_source"], g["deg_target"] = s, t
    # a pose that is NOT the identity (under the identity both summation orders return the point itself)
    Tdeg = scene.make_transform(np.array([0.013, -0.007, 0.004]), np.deg2rad(0.11), np.deg2rad(0.02), np.deg2rad(-0.03))
    g["deg_T"] = Tdeg
    for sfx, fl in XF:
        kw = dict(max_correspondence_distance=2.0, flags=fl)
        r = R.RefAPDGICP(R.default_params(**kw))
        n = O.FastAPDGICP(np_params(**kw))
        for o in (r, n):
            o.setInputSource(s)
            o.setInputTarget(t)
        n.source_covs, n.target_covs = O.calculate_covariances(s), O.calculate_covariances(t)
        for ptag, Tp in (("", np.eye(4)), ("_moved", Tdeg)):
            cr, Hr, br = r.linearize(Tp)
            cn, Hn, bn = n.linearize(Tp)
            corr, sqd = r.correspondences()
            assert np.array_equal(corr, n.correspondences) and np.array_equal(sqd, n.sq_distances)
            assert corr[2] == -1 and corr[0] == 0 and corr[1] == 1
            worst["deg"] = max(worst.get("deg", 0), rel(Hr, Hn), rel(br, bn))
            assert rel(Hr, Hn) < 1e-8 and rel(br, bn) < 1e-8   # (1 / cos(AoA) near the +x axis amplifies the last bits of the fp64 trigonometry)
            k_ = f"deg{ptag}{sfx}"
            g[f"{k_}_corr"], g[f"{k_}_sqd"], g[f"{k_}_H"], g[f"{k_}_b"], g[f"{k_}_cost"] = corr, sqd, Hr, br, cr
            g[f"{k_}_maha128"] = r.mahalanobis()[:128]

    # (b) LM rejection (rho < 0, L:156-164) and the "lm not converged" failure path (L:71-74,172).
    #     On radar-range data LM never rejects a step (scanned: thousands of iterations), so these use
    #     far-range clouds (lever arm 500-3000 m) where the so(3) linearisation overshoots.
    found = False
    for trial in range(200):
        rng = np.random.default_rng(4242 + trial)
        nn_ = 256
        rr = rng.choice([200, 500, 1000, 3000])
        t = (rng.normal(size=(nn_, 3)) * [5, 5, 1.0] + [rr, 0, 0]).astype(np.float32)
        yaw = rng.uniform(0.2, 3.0)
        Tt = scene.make_transform(rng.normal(size=3) * 0.5, np.deg2rad(yaw), 0, 0)
        Ti = np.linalg.inv(Tt)
        s = ((t.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]) + rng.normal(size=(nn_, 3)) * 0.02).astype(np.float32)
        ok = True
        for sfx, fl in XF:
            r = R.RefAPDGICP(R.default_params(flags=fl))
            r.setInputSource(s)
            r.setInputTarget(t)
            r.align(None)
            ok &= not (r.n_compute_error == r.n_linearize or not r.converged)
        if not ok:
            continue
        stash = {}
        # "fail": same clouds, lm_max_iterations=1 -> the first rejected step ends the run (L:71-74,172)
        for tag, kw0, sfx, fl in [(t_, k_, s_, f) for t_, k_ in (("rej", {}), ("fail", dict(lm_max_iterations=1))) for s_, f in XF]:
            kw = dict(kw0, flags=fl)
            base_tag, tag = tag, tag + sfx
            r = R.RefAPDGICP(R.default_params(**kw))
            n = O.FastAPDGICP(np_params(**kw))
            for o in (r, n):
                o.setInputSource(s)
                o.setInputTarget(t)
            Tr, Tn = r.align(None), n.align(None)
            te, re_ = scene.pose_error(Tr, Tn)
            same = (r.converged, r.nr_iterations, r.n_linearize, r.n_compute_error) == (
                n.converged, n.nr_iterations, n.trace.n_linearize, n.trace.n_compute_error)
            if not same or te > 1e-6 * rr or re_ > 1e-6:
                ok = False
                break
            if base_tag == "fail" and (r.converged or r.nr_iterations >= 63):
                ok = False
                break
            stash[f"{tag}_T"] = Tr
            stash[f"{tag}_info"] = np.array([r.converged, r.nr_iterations, r.n_linearize, r.n_compute_error], dtype=np.int32)
            try:
                for k2, v2 in check_traces(tag, r, n).items():
                    stash[f"{tag}_trace_{k2}"] = v2
            except AssertionError:
                ok = False
                break
            stash[f"{tag}_msg"] = (tag, "trial", trial, "range", rr, "converged", r.converged, "iters", r.nr_iterations,
                                   "lin", r.n_linearize, "err", r.n_compute_error, "rho<0:", int((np.array(n.trace.rhos) < 0).sum()))
        if not ok:
            continue
        for k_, v_ in stash.items():
            if k_.endswith("_msg"):
                print(*v_)
            else:
                g[k_] = v_
        g["rej_source"], g["rej_target"] = s, t
        found = True
        break
    assert found, "no agreeing LM-rejection case found"

    print("largest differences between the two restatements:", {k_: float(f"{v_:.3g}") for k_, v_ in worst.items()})
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB,", len(g), "arrays")


if __name__ == "__main__":
    main()
