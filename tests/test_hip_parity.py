"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the committed golden
vectors and against the CPU oracle on seeded inputs.

Bars: correspondences / iteration counts / converged flags exact; fp32 squared distances bit-exact;
covariances 1e-10; M, H, b, cost 1e-10 relative (HB_TOL: kernels and checker evaluate the three fp32 angles of the sensor
model with the same atan2f, include/apd_atan2f.h = the C library's algorithm, tests/test_atan2f.py; what is left is the order
of the fp64 sums, fused multiply-adds and the last bits of sin / cos); per-iteration optimiser traces against the golden
traces; final transforms <= 1e-3 m and <= 1e-4 rad (north_star) -- asserted much tighter where the run is well conditioned.
"""
import importlib
import os

import numpy as np
import pytest

import ref as R
from conftest import rel_err
from trace_util import golden_trace, trace_close

pytestmark = pytest.mark.gpu

LAUNCH = dict(max_correspondence_distance=2.0, transformation_epsilon=0.1, azimuth_variance_deg=1.0)
REGS = (("none", 0), ("min_eig", 1), ("norm_min_eig", 2), ("plane", 3), ("frobenius", 4))
T_TOL, R_TOL = 1e-3, 1e-4
HB_TOL = 1e-10   # M, H, b, cost against the oracle / the golden vectors (relative); until round 5: 5e-6 (the device library's atan2f)
# The fp32 summation order of T * p (A:149) is Eigen's: pairwise for Eigen >= 3.3 (default), a linear chain for Eigen 3.2
# (APDGICP_FLAG_XF_LINEAR_CHAIN).  Golden keys of the second carry the suffix; oracle and product take the same flag.
XF = (pytest.param("", 0, id="xf_pairwise"), pytest.param("_xflin", 2, id="xf_linear"))


@pytest.fixture(scope="module")
def reg():
    import __graft_entry__ as g
    g.build()
    return importlib.import_module("riv-slam_amd.registration")


def both(reg, src, tgt, **kw):
    g = reg.FastAPDGICP(reg.default_params(**kw))
    o = R.RefAPDGICP(R.default_params(**kw))
    for x in (g, o):
        x.setInputSource(src)
        x.setInputTarget(tgt)
    return g, o


def info_of(g):
    r = g.result
    return [int(r.converged), int(r.iterations), int(r.n_linearize), int(r.n_compute_error)]


# ------------------------------------------------------------------ covariances (a6)
@pytest.mark.parametrize("name,mode", REGS)
def test_cov_golden(reg, golden, name, mode):
    g = reg.FastAPDGICP(reg.default_params(regularization=mode))
    g.setInputSource(golden["cov_cloud"])
    c = g.getSourceCovariances()
    assert np.all(c[:, 3, :] == 0) and np.all(c[:, :, 3] == 0)
    assert np.abs(c[:, :3, :3] - golden[f"cov_{name}"]).max() <= 1e-10


def test_cov_vs_oracle_2k_and_ragged(reg, scene):
    for n in (20, 21, 127, 129, 1000, 2048, 2500):
        src, _, _, _ = scene.make_pair(n, 32, scene.pair_seed(7, n), "odometry")
        g = reg.FastAPDGICP()
        g.setInputSource(src)
        o = R.RefAPDGICP()
        o.setInputSource(src)
        assert np.abs(g.getSourceCovariances()[:, :3, :3] - o.covariances("source")).max() <= 1e-10, n


def test_cov_k_values(reg, golden):
    cloud = golden["lin_source"][:700]
    for k in (5, 10, 32, 33, 48, 64):   # (beyond 32: the brute-force kernel with 128-entry lists)
        g = reg.FastAPDGICP(reg.default_params(k_correspondences=k, regularization=0))
        g.setInputSource(cloud)
        o = R.RefAPDGICP(R.default_params(k_correspondences=k, regularization=0))
        o.setInputSource(cloud)
        assert np.abs(g.getSourceCovariances()[:, :3, :3] - o.covariances("source")).max() <= 1e-10, k


@pytest.mark.parametrize("n,k", ((16384, 33), (16384, 64), (30000, 48), (110_000, 64), (64, 64), (100, 40)))
def test_cov_k_beyond_32_at_scan_and_submap_sizes(reg, scene, n, k):
    """32 < k <= 64 goes through the brute-force covariance kernel with 128-entry lists, whose bound comes from 64 class minima.
    (it used to start unbounded and ran into its restart guard -- error flag, garbage -- from ~16k points on.)"""
    src, _, _, _ = scene.make_pair(n, 32, scene.pair_seed(7, n + k), "odometry")
    g = reg.FastAPDGICP(reg.default_params(k_correspondences=k, regularization=0))
    g.setInputSource(src)
    o = R.RefAPDGICP(R.default_params(k_correspondences=k, regularization=0))
    o.setInputSource(src)
    assert np.abs(g.getSourceCovariances()[:, :3, :3] - o.covariances("source")).max() <= 1e-10


@pytest.mark.parametrize("n,k,mode", ((4000, 65, 0), (4000, 128, 3), (8192, 100, 3), (300, 200, 1), (66, 66, 0), (1001, 1000, 2)))
def test_cov_any_k_above_64(reg, scene, n, k, mode):
    """k > 64 (the reference takes any k, A:45-47): the selection kernel -- the k-th key by bisection, no lists -- against the oracle;
    with duplicated points the (distance, original index) rule decides which copies belong to the k."""
    src, _, _, _ = scene.make_pair(n, 32, scene.pair_seed(7, 3 * n + k), "odometry")
    if n == 4000:
        src[1::3] = src[0:-1:3][: len(src[1::3])]      # a third of the points twice: ties at the k-th distance
    g = reg.FastAPDGICP(reg.default_params(k_correspondences=k, regularization=mode))
    g.setInputSource(src)
    o = R.RefAPDGICP(R.default_params(k_correspondences=k, regularization=mode))
    o.setInputSource(src)
    co = o.covariances("source")
    assert np.abs(g.getSourceCovariances()[:, :3, :3] - co).max() <= 1e-10 * max(1.0, np.abs(co).max())
    if n == 8192:   # and a whole registration with it
        s2, t2, _, guess = scene.make_pair(2000, 2500, scene.pair_seed(7, 99), "odometry")
        g2, o2 = both(reg, s2, t2, k_correspondences=k, max_correspondence_distance=2.0)
        T, To = g2.align(guess), o2.align(guess)
        te, re_ = scene.pose_error(To, T)
        assert te <= T_TOL and re_ <= R_TOL and info_of(g2) == [int(o2.converged), o2.nr_iterations, o2.n_linearize, o2.n_compute_error]


@pytest.mark.parametrize("n", (2000, 8192))   # (8192: host clouds read by the sort from pinned memory, bounding box from the host)
def test_non_finite_points_fail_loudly(reg, scene, n):
    """The preprocessing nodelet removes NaNs before registration (preprocessing_nodelet.cpp); a cloud that still carries
    non-finite coordinates is reported as an error -- no hang, no silent garbage -- and the handle stays usable."""
    src, tgt, _, guess = scene.make_pair(n, n, 5, "odometry")
    bad_s, bad_t = src.copy(), tgt.copy()
    bad_s[[3, 500, 1999]] = np.nan
    bad_t[[7, 900]] = np.inf
    bad_t[1000, 1] = -np.inf
    g = reg.FastAPDGICP(reg.default_params(max_correspondence_distance=2.0))
    g.setInputSource(bad_s)
    g.setInputTarget(bad_t)
    with pytest.raises(Exception, match="non-finite"):
        g.align(guess)
    g.setInputSource(src)
    g.setInputTarget(tgt)
    T = g.align(guess)
    fresh = reg.FastAPDGICP(reg.default_params(max_correspondence_distance=2.0))
    fresh.setInputSource(src)
    fresh.setInputTarget(tgt)
    assert np.array_equal(T, fresh.align(guess))


def test_degenerate_neighbourhoods_get_a_positive_definite_regularised_covariance(reg):
    """Collinear / coplanar / duplicated neighbourhoods have a rank-deficient covariance.  The reference rebuilds
    svd.matrixU() * diag * svd.matrixV()^T (fast_apdgicp_impl.hpp:337-357), where JacobiSVD may return U and V columns of opposite
    sign for a ZERO singular value -- an indefinite matrix; this implementation (and the restatement it is checked against) uses
    the symmetric eigen-decomposition U diag U^T, which is what the formula means for a PSD input.  Pinned here: PLANE gives
    eigenvalues exactly {1, 1, 1e-3}, MIN_EIG >= 1e-3, all symmetric positive definite (deviations)."""
    rng = np.random.default_rng(5)
    line = np.stack([np.linspace(0, 6.3, 64), np.zeros(64), np.zeros(64)], axis=1)
    plane = np.stack([rng.uniform(10, 12, 64), rng.uniform(-1, 1, 64), np.full(64, 0.5)], axis=1)
    dup = np.tile(np.array([[20.0, 3.0, 1.0]]), (40, 1))
    for mode, lo in ((reg.REG_PLANE, 1e-3), (reg.REG_MIN_EIG, 1e-3), (reg.REG_NORMALIZED_MIN_EIG, 1e-3)):
        # (an all-duplicates neighbourhood has a ZERO covariance: NORMALIZED_MIN_EIG divides by its largest eigenvalue, 0 / 0, in
        # the reference as well, so that mode is pinned on the line and the plane only)
        pts = np.concatenate([line, plane] + ([] if mode == reg.REG_NORMALIZED_MIN_EIG else [dup])).astype(np.float32)
        g = reg.FastAPDGICP(reg.default_params(regularization=mode, k_correspondences=10))
        g.setInputSource(pts)
        cov = g.getSourceCovariances()[:, :3, :3]
        o = R.RefAPDGICP(R.default_params(regularization=mode, k_correspondences=10))
        o.setInputSource(pts)
        assert np.abs(cov - o.covariances("source")).max() < 1e-9
        assert np.abs(cov - cov.transpose(0, 2, 1)).max() < 1e-15
        ev = np.linalg.eigvalsh(cov)
        assert ev.min() > lo * (1 - 1e-9), (mode, ev.min())
        if mode == reg.REG_PLANE:
            assert np.abs(ev - np.array([1e-3, 1.0, 1.0])).max() < 1e-9


def test_cov_duplicate_points(reg):
    """Many identical points: ties are resolved by index, the selection must still be exact."""
    rng = np.random.default_rng(5)
    base = rng.uniform(0, 10, size=(40, 3)).astype(np.float32)
    cloud = np.repeat(base, 8, axis=0)  # every point 8 times -> 320 points, massive ties
    g = reg.FastAPDGICP(reg.default_params(regularization=0))
    g.setInputSource(cloud)
    o = R.RefAPDGICP(R.default_params(regularization=0))
    o.setInputSource(cloud)
    assert np.abs(g.getSourceCovariances()[:, :3, :3] - o.covariances("source")).max() <= 1e-10


def test_cov_lists_overflow_many_times_bitwise_vs_brute_force(reg):
    """The four-lanes-per-query k-NN (>= 100 000 points per launch) keeps at most 60 candidates per query and tightens a full
    list to a key of rank k .. 40 by bisection.  Dense clusters (hundreds of points inside a query's first bound), exact
    duplicates (equal distances, ordered by index) and a sparse background make lists overflow again and again; the
    covariances must stay those of the brute-force kernel, bit for bit."""
    rng = np.random.default_rng(11)
    centres = rng.uniform(-40, 40, size=(300, 3)).astype(np.float32)
    clusters = (centres[:, None, :] + rng.normal(0, 0.02, size=(300, 250, 3)).astype(np.float32)).reshape(-1, 3)   # 75 000
    dup = np.repeat(rng.uniform(-40, 40, size=(1500, 3)).astype(np.float32), 10, axis=0)                           # 15 000, each point 10 times
    back = rng.uniform(-60, 60, size=(20_000, 3)).astype(np.float32)
    cloud = np.concatenate([clusters, dup, back])
    cloud = cloud[rng.permutation(len(cloud))]
    assert len(cloud) >= 100_000
    for k in (20, 32):
        a, b = _fresh(reg, "pruned", regularization=0, k_correspondences=k), _fresh(reg, "brute", regularization=0, k_correspondences=k)
        a.setInputSource(cloud)
        b.setInputSource(cloud)
        assert np.array_equal(a.getSourceCovariances(), b.getSourceCovariances()), k


# ------------------------------------------------------------------ linearize / compute_error (a7-a9)
@pytest.mark.parametrize("sfx,flags", XF)
@pytest.mark.parametrize("tag,kw", (("default", {}), ("launch", LAUNCH)))
def test_linearize_golden(reg, golden, tag, kw, sfx, flags):
    g = reg.FastAPDGICP(reg.default_params(flags=flags, **kw))
    tag = tag + sfx
    g.setInputSource(golden["lin_source"])
    g.setInputTarget(golden["lin_target"])
    for k in range(3):
        cost, H, b = g.linearize(golden[f"lin_{tag}_{k}_T"])
        corr, sqd = g.correspondences()
        assert np.array_equal(corr, golden[f"lin_{tag}_{k}_corr"])
        assert np.array_equal(sqd.view(np.uint32), golden[f"lin_{tag}_{k}_sqd"].view(np.uint32))
        assert rel_err(H, golden[f"lin_{tag}_{k}_H"]) < HB_TOL
        assert rel_err(b, golden[f"lin_{tag}_{k}_b"]) < HB_TOL
        assert abs(cost - golden[f"lin_{tag}_{k}_cost"]) < HB_TOL * cost
        assert np.allclose(H, H.T)
        assert rel_err(g.mahalanobis()[:128], golden[f"lin_{tag}_{k}_maha128"]) < HB_TOL
        err = g.compute_error(golden[f"lin_{tag}_{k}_errT"])
        assert abs(err - golden[f"lin_{tag}_{k}_err"]) < HB_TOL * err
        cost_only, _, _ = g.linearize(golden[f"lin_{tag}_{k}_T"], want_Hb=False)
        assert abs(cost_only - cost) <= 1e-12 * cost


def test_linearize_with_injected_covariances_is_tight(reg, golden):
    """With the oracle's covariances injected (setSourceCovariances, A:111-118) only the per-point
    kernel differs; compare_error == linearize cost at the same pose (frozen state)."""
    g = reg.FastAPDGICP(reg.default_params(**LAUNCH))
    g.setInputSource(golden["lin_source"])
    g.setInputTarget(golden["lin_target"])
    g.setSourceCovariances(golden["lin_source_cov"])
    g.setTargetCovariances(golden["lin_target_cov"])
    T = golden["lin_launch_1_T"]
    cost, H, b = g.linearize(T)
    assert rel_err(H, golden["lin_launch_1_H"]) < HB_TOL
    assert abs(g.compute_error(T) - cost) <= 1e-12 * cost
    assert np.abs(g.getSourceCovariances()[:, :3, :3] - golden["lin_source_cov"]).max() <= 1e-15  # upper triangle is stored


@pytest.mark.parametrize("sfx,flags", XF)
def test_degenerate_golden(reg, golden, sfx, flags):
    g = reg.FastAPDGICP(reg.default_params(max_correspondence_distance=2.0, flags=flags))
    g.setInputSource(golden["deg_source"])
    g.setInputTarget(golden["deg_target"])
    for ptag, T in (("", np.eye(4)), ("_moved", golden["deg_T"])):  # (under the identity both orders return the point itself)
        k = f"deg{ptag}{sfx}"
        cost, H, b = g.linearize(T)
        corr, sqd = g.correspondences()
        assert np.array_equal(corr, golden[f"{k}_corr"]) and corr[2] == -1
        assert np.array_equal(sqd.view(np.uint32), golden[f"{k}_sqd"].view(np.uint32))
        assert rel_err(H, golden[f"{k}_H"]) < HB_TOL and rel_err(b, golden[f"{k}_b"]) < HB_TOL
        M = g.mahalanobis()
        assert np.all(M[2] == 0)
        # the +x-axis points carry the APD blow-up (1 / cos(AoA) with AoA an ulp from pi / 2 amplifies the last bits of the fp64 cosine)
        assert rel_err(M[:2], golden[f"{k}_maha128"][:2]) < 1e-6
        assert rel_err(M[3:128], golden[f"{k}_maha128"][3:128]) < HB_TOL


def test_the_two_transform_orders_are_two_different_searches(reg, golden):
    """The flag reaches every kernel that transforms a point: at a non-trivial pose the fp32 distances of the two modes differ in
    the last bit for many points (and each mode matches ITS golden vector bit for bit, test_linearize_golden)."""
    out = []
    for flags in (0, reg.FLAG_XF_LINEAR_CHAIN):
        g = reg.FastAPDGICP(reg.default_params(flags=flags, **LAUNCH))
        g.setInputSource(golden["lin_source"])
        g.setInputTarget(golden["lin_target"])
        g.linearize(golden["lin_launch_1_T"])
        out.append(g.correspondences()[1].view(np.uint32).copy())
    want = int((golden["lin_launch_1_sqd"].view(np.uint32) != golden["lin_launch_xflin_1_sqd"].view(np.uint32)).sum())
    assert want > 100 and int((out[0] != out[1]).sum()) == want


def test_all_unmatched_is_not_an_error(reg, golden):
    """No correspondence inside the gate: H = 0, d = 0, the reference converges at once (L:156-159 / rho = NaN)."""
    src = golden["lin_source"][:256] + np.float32(1000.0)
    kw = dict(max_correspondence_distance=1.0)
    g, o = both(reg, src, golden["lin_target"][:256], **kw)
    T, To = g.align(None), o.align(None)
    assert info_of(g) == [int(o.converged), o.nr_iterations, o.n_linearize, o.n_compute_error]
    assert np.array_equal(T, To) and g.result.n_matched == 0


# ------------------------------------------------------------------ the optimiser (a11-a14)
RUNS = {
    "lm_default": {},
    "lm_launch": LAUNCH,
    "gn20": dict(optimizer=1, max_iterations=20, transformation_epsilon=1e-300, rotation_epsilon=1e-300),
    "lm_loop": dict(max_correspondence_distance=2.5),
}


@pytest.mark.parametrize("sfx,flags", XF)
@pytest.mark.parametrize("host_loop", (False, True))
@pytest.mark.parametrize("tag", list(RUNS))
def test_align_golden(reg, golden, scene, tag, host_loop, sfx, flags):
    pre = "lm_loop" if tag == "lm_loop" else "lin"
    g = reg.FastAPDGICP(reg.default_params(flags=flags, **RUNS[tag]))
    g.setInputSource(golden[f"{pre}_source"])
    g.setInputTarget(golden[f"{pre}_target"])
    T = g.align(golden[f"{pre}_guess"], host_loop=host_loop)
    tag = tag + sfx
    assert info_of(g) == list(golden[f"{tag}_info"])
    te, re_ = scene.pose_error(golden[f"{tag}_T"], T)
    assert te <= T_TOL and re_ <= R_TOL
    print(tag, host_loop, 'pose diff vs golden', te, re_)
    assert te <= 1e-9 and re_ <= 1e-10, (te, re_)      # (measured: <= 1.2e-16 m / 2.5e-18 rad, profiles/r05_trace_c4.log)
    assert rel_err(g.getFinalHessian(), golden[f"{tag}_final_hessian"]) < 1e-10
    assert g.hasConverged() == bool(golden[f"{tag}_info"][0])


@pytest.mark.parametrize("sfx,flags", XF)
@pytest.mark.parametrize("host_loop", (False, True))
@pytest.mark.parametrize("tag", list(RUNS) + ["rej", "fail"])
def test_optimiser_trace_golden(reg, golden, tag, host_loop, sfx, flags):
    """SURVEY 8(c) KAT-lm: the trajectory, not only its end -- per Levenberg-Marquardt trial the lambda the step was solved with,
    its gain ratio rho and the two costs it compares, per outer iteration the pose behind it (L:131-169) -- of the device state
    machine (a debug ring written by the last block of k_linearize / k_error) and of the host-driven loop, against the golden
    traces of the two CPU restatements.  Bars (tests/trace_util.py): costs 1e-11 relative, rho and lambda 1e-11 times the
    cancellation y0 / |y0 - yi| of the gain ratio, poses 1e-9 m (measured: 1.3e-15 relative on radar-range clouds, 1.2e-12 on the
    far-range rejection clouds, poses 3e-15 m / 1e-12 m; profiles/r05_trace_c4.log)."""
    kw = dict(RUNS.get(tag, {}), **(dict(lm_max_iterations=1) if tag == "fail" else {}))
    pre = "lm_loop" if tag == "lm_loop" else "rej" if tag in ("rej", "fail") else "lin"
    g = reg.FastAPDGICP(reg.default_params(flags=flags, **kw))
    g.setTrace(True)
    g.setInputSource(golden[f"{pre}_source"])
    g.setInputTarget(golden[f"{pre}_target"])
    g.align(None if pre == "rej" else golden[f"{pre}_guess"], host_loop=host_loop)
    assert info_of(g) == list(golden[f"{tag}{sfx}_info"])
    tr, want = g.trace(), golden_trace(golden, tag + sfx)
    assert len(tr["rho"]) == g.result.n_compute_error == len(want["rho"])
    if tag in ("rej", "fail"):
        assert np.array_equal(tr["rho"] < 0, want["rho"] < 0) and (tr["rho"] < 0).any()   # the same trials are rejected (L:156)
    d = trace_close(tr, want, tol_cost=1e-11, tol_pose=1e-9)
    print(tag + sfx, "host loop" if host_loop else "device loop", "normalised trace differences", d)
    assert max(d.values()) < 1.0, d


@pytest.mark.parametrize("sfx,flags", XF)
@pytest.mark.parametrize("tag,kw", (("rej", {}), ("fail", dict(lm_max_iterations=1))))
def test_lm_rejection_and_failure_paths(reg, golden, scene, tag, kw, sfx, flags):
    """rho < 0 -> lambda *= nu (L:156-164) and the 'lm not converged' exit (L:71-74,172)."""
    g = reg.FastAPDGICP(reg.default_params(flags=flags, **kw))
    g.setInputSource(golden["rej_source"])
    g.setInputTarget(golden["rej_target"])
    for host_loop in (False, True):
        T = g.align(None, host_loop=host_loop)
        assert info_of(g) == list(golden[f"{tag}{sfx}_info"])
        assert g.result.n_compute_error > g.result.n_linearize or tag == "fail"
        te, re_ = scene.pose_error(golden[f"{tag}{sfx}_T"], T)
        assert te <= T_TOL and re_ <= R_TOL
        assert bool(g.result.lm_failed) == (tag == "fail")


def test_max_iterations_zero_returns_guess(reg, golden):
    g = reg.FastAPDGICP(reg.default_params(max_iterations=0))
    g.setInputSource(golden["lin_source"][:512])
    g.setInputTarget(golden["lin_target"][:512])
    T = g.align(golden["lin_guess"])
    assert np.array_equal(T[:3], golden["lin_guess"][:3]) and not g.hasConverged() and g.result.n_linearize == 0


@pytest.mark.parametrize("sfx,flags", XF)
def test_align_seeded_pairs_vs_oracle(reg, scene, sfx, flags):
    """Seeded pairs the golden file does not hold: odometry + loop closure, launch parameters."""
    for idx, (kind, n, m) in enumerate((("odometry", 3000, 2500), ("loop", 1500, 4097), ("odometry", 4096, 4096))):
        src, tgt, _, guess = scene.make_pair(n, m, scene.pair_seed(3, idx), kind)
        kw = dict(max_correspondence_distance=2.5, azimuth_variance_deg=1.0, flags=flags)
        g, o = both(reg, src, tgt, **kw)
        T, To = g.align(guess), o.align(guess)
        assert info_of(g) == [int(o.converged), o.nr_iterations, o.n_linearize, o.n_compute_error]
        te, re_ = scene.pose_error(To, T)
        assert te <= T_TOL and re_ <= R_TOL, (kind, te, re_)
        cg, _ = g.correspondences()
        co, _ = o.correspondences()
        assert np.array_equal(cg, co)


@pytest.mark.parametrize("sfx,flags", XF)
def test_full_size_8k_pair(reg, scene, sfx, flags):
    """BASELINE configs[1]: 8k x 8k, GN-20; correspondences at the guess bit-exact, final pose within the
    north-star tolerance, plus size-independent properties."""
    src, tgt, _, guess = scene.make_pair(8192, 8192, scene.pair_seed(2, 0), "odometry")
    kw = dict(optimizer=1, max_iterations=20, transformation_epsilon=1e-300, rotation_epsilon=1e-300,
              max_correspondence_distance=2.0, azimuth_variance_deg=1.0, flags=flags)
    g, o = both(reg, src, tgt, **kw)
    c1, H1, b1 = g.linearize(guess.astype(np.float64))
    c2, H2, b2 = o.linearize(guess.astype(np.float64))
    cg, sg = g.correspondences()
    co, so = o.correspondences()
    assert np.array_equal(cg, co) and np.array_equal(sg.view(np.uint32), so.view(np.uint32))
    assert rel_err(H1, H2) < HB_TOL and rel_err(b1, b2) < HB_TOL and abs(c1 - c2) < HB_TOL * c2
    T, To = g.align(guess), o.align(guess)
    assert g.result.n_linearize == 20 == o.n_linearize
    te, re_ = scene.pose_error(To, T)
    assert te <= T_TOL and re_ <= R_TOL, (te, re_)
    # property: registering a cloud against itself from identity is a fixed point
    g.setInputSource(tgt)
    Ti = g.align(None)
    assert np.abs(Ti - np.eye(4)).max() < 1e-6
    # property: linearize is invariant to the order of the target points except for the indices
    perm = np.random.default_rng(0).permutation(len(tgt))
    g.setInputSource(src)
    g.setInputTarget(tgt[perm])
    c3, H3, b3 = g.linearize(guess.astype(np.float64))
    cp, sp = g.correspondences()
    assert np.array_equal(sp.view(np.uint32), sg.view(np.uint32))
    ok = cp >= 0
    assert np.array_equal(ok, cg >= 0)
    same_point = np.all(tgt[perm][cp[ok]] == tgt[cg[ok]], axis=1)
    assert same_point.mean() > 0.999  # exact-tie targets may swap
    assert rel_err(H3, H1) < 1e-6


# ------------------------------------------------------------------ object semantics (a3-a5)
def test_caching_tokens_swap_and_clear(reg, golden, scene):
    src, tgt, guess = golden["lin_source"], golden["lin_target"], golden["lin_guess"]
    g = reg.FastAPDGICP(reg.default_params(**LAUNCH))
    g.setInputSource(src, token=11)
    g.setInputTarget(tgt, token=22)
    T1 = g.align(guess)
    g.setInputTarget(np.zeros_like(tgt), token=22)   # same token: pointer-equality early return (A:102-104)
    T2 = g.align(guess)
    assert np.array_equal(T1, T2)
    g.swapSourceAndTarget()                          # A:68-75
    Tinv = g.align(np.linalg.inv(guess.astype(np.float64)).astype(np.float32))
    o = R.RefAPDGICP(R.default_params(**LAUNCH))
    o.setInputSource(tgt)
    o.setInputTarget(src)
    To = o.align(np.linalg.inv(guess.astype(np.float64)).astype(np.float32))
    te, re_ = scene.pose_error(To, Tinv)
    assert te <= T_TOL and re_ <= R_TOL
    g.clearSource()                                  # A:78-81
    with pytest.raises(reg.ApdgicpError) as e:
        g.align(guess)
    assert e.value.code == -3


@pytest.mark.parametrize("stride_floats", (3, 4, 8))
def test_point_strides(reg, golden, stride_floats):
    """packed xyz, float4 and the 32-byte pcl::PointXYZI layout (x,y,z,1,intensity,pad)."""
    src, tgt = golden["lin_source"][:999], golden["lin_target"][:1001]

    def widen(a):
        w = np.full((len(a), stride_floats), 7.0, dtype=np.float32)
        w[:, :3] = a
        return w
    g = reg.FastAPDGICP(reg.default_params(**LAUNCH))
    g.setInputSource(widen(src))
    g.setInputTarget(widen(tgt))
    h = reg.FastAPDGICP(reg.default_params(**LAUNCH))
    h.setInputSource(src)
    h.setInputTarget(tgt)
    a, b = g.linearize(np.eye(4)), h.linearize(np.eye(4))
    assert a[0] == b[0] and np.array_equal(a[1], b[1])


def test_device_pointer_input_and_output_cloud(reg, golden):
    import torch
    src, tgt, guess = golden["lin_source"], golden["lin_target"], golden["lin_guess"]
    g = reg.FastAPDGICP(reg.default_params(**LAUNCH))
    g.setInputSource(torch.from_numpy(src).cuda())
    g.setInputTarget(torch.from_numpy(tgt).cuda())
    out = g.align(guess, want_output=True)            # pcl::transformPointCloud, L:79
    T = g.getFinalTransformation()
    want = src @ T[:3, :3].T + T[:3, 3]
    assert np.abs(out - want).max() < 1e-4
    h = reg.FastAPDGICP(reg.default_params(**LAUNCH))
    h.setInputSource(src)
    h.setInputTarget(tgt)
    assert np.array_equal(h.align(guess), T)


def test_errors_are_codes_not_crashes(reg, golden):
    g = reg.FastAPDGICP()
    with pytest.raises(reg.ApdgicpError) as e:
        g.align(None)
    assert e.value.code == -3
    g.setInputSource(golden["lin_source"][:10])   # fewer than k points
    g.setInputTarget(golden["lin_target"][:100])
    with pytest.raises(reg.ApdgicpError) as e:
        g.align(None)
    assert e.value.code == -4
    with pytest.raises(reg.ApdgicpError) as e:
        g.setCorrespondenceRandomness(0)
    assert e.value.code == -1
    g.params.k_correspondences = 20
    g.setCorrespondenceRandomness(101)            # any k is accepted (A:45-47) ...
    with pytest.raises(reg.ApdgicpError) as e:    # ... and a cloud with fewer points than k is an error code, as for k = 20
        g.align(None)
    assert e.value.code == -4
    g.setCorrespondenceRandomness(20)
    with pytest.raises(reg.ApdgicpError):
        g.setRegularizationMethod(9)
    with pytest.raises(reg.ApdgicpError):
        g.compute_error(np.eye(4))
    g.params.regularization = 3
    with pytest.raises(reg.ApdgicpError) as e:   # an unknown flag bit is refused, not ignored
        g.set_params(reg.default_params(flags=16))
    assert e.value.code == -1
    with pytest.raises(reg.ApdgicpError) as e:   # the two opt-in arithmetic modes are exclusive
        g.set_params(reg.default_params(flags=reg.FLAG_FP32_POINT_MATH | reg.FLAG_ALGEBRAIC_APD))
    assert e.value.code == -1
    g.set_params(reg.default_params())
    g.setTransformOrder(True)
    assert g.params.flags == reg.FLAG_XF_LINEAR_CHAIN
    g.setTransformOrder(False)
    assert g.params.flags == 0


def test_two_handles_two_threads(reg, golden):
    """Several registration objects coexist in one process (three in the nodelet manager)."""
    import threading
    src, tgt, guess = golden["lin_source"], golden["lin_target"], golden["lin_guess"]
    outs = {}

    def work(i):
        g = reg.FastAPDGICP(reg.default_params(**LAUNCH))
        g.setInputSource(src)
        g.setInputTarget(tgt)
        for _ in range(3):
            outs[i] = g.align(guess)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_fitness_score(reg, golden):
    import apdgicp_np as O
    src, tgt, guess = golden["lin_source"], golden["lin_target"], golden["lin_guess"]
    g = reg.FastAPDGICP(reg.default_params(**LAUNCH))
    g.setInputSource(src)
    g.setInputTarget(tgt)
    T = g.align(guess)
    pt = O.transform_points_f32(T.astype(np.float64), src)
    _, sq = O.nn1(pt, tgt)
    for max_range in (np.finfo(np.float64).max, 4.0, 0.25):
        sel = sq.astype(np.float64) <= max_range
        want = sq[sel].astype(np.float64).mean()
        got = g.getFitnessScore(max_range)
        assert abs(got - want) < 1e-9 * want and g.last_inliers == sel.sum()


def test_inlier_fraction_is_strict(reg, golden):
    """ScanMatchingStatus.inlier_fraction (scan_matching_odometry_nodelet.cpp:701-712): `sq < d*d`, strictly, count / size in float."""
    import apdgicp_np as O
    src, tgt, guess = golden["lin_source"], golden["lin_target"], golden["lin_guess"]
    g = reg.FastAPDGICP(reg.default_params(**LAUNCH))
    g.setInputSource(src)
    g.setInputTarget(tgt)
    T = g.align(guess)
    _, sq = O.nn1(O.transform_points_f32(T.astype(np.float64), src), tgt)
    exact = float(np.sqrt(np.float64(np.sort(sq)[len(sq) // 2])))   # a threshold that IS some point's distance: `<` vs `<=` differ
    for d in (0.5, 0.1, exact, 1e-9):
        want = int((sq.astype(np.float64) < d * d).sum())
        got = g.inlierFraction(d)
        assert g.last_inliers == want and got == float(np.float32(want) / np.float32(len(src))), d
    assert g.getFitnessScore(exact * exact) > 0 and g.last_inliers >= int((sq.astype(np.float64) < exact * exact).sum())


def test_temporary_device_tensors_are_safe(reg, scene):
    """A device cloud handed over as a temporary must stay alive, and the handle must wait for the stream that produced it:
    torch's caching allocator would otherwise recycle the block under the queued pack kernel."""
    import torch
    s, t, _, guess = scene.make_pair(4096, 4096, scene.pair_seed(7, 1), "odometry")
    want = None
    for rep_ in range(6):
        g = reg.FastAPDGICP(reg.default_params(**LAUNCH))
        big = torch.zeros(64 << 20, device="cuda")          # keeps torch's stream busy in front of the producer
        for _ in range(4):
            big.add_(1.0)
        g.setInputSource((torch.from_numpy(s).cuda() + big[: s.size].view(s.shape) * 0.0))   # produced late on torch's stream, temporary
        junk1 = torch.full(s.shape, 1e6, device="cuda")     # same size: would reuse the block if it had been released
        g.setInputTarget(torch.from_numpy(t).cuda() * 1.0)
        junk2 = torch.full(t.shape, -1e6, device="cuda")
        T = g.align(guess)
        info = (info_of(g), T.tobytes())
        want = want or info
        assert info == want
        del junk1, junk2, big
    h = reg.FastAPDGICP(reg.default_params(**LAUNCH))
    h.setInputSource(s), h.setInputTarget(t)
    assert h.align(guess).tobytes() == want[1] and info_of(h) == want[0]
    # the batch path: temporaries across an enqueue / collect pipeline
    b = reg.BatchAPDGICP(reg.default_params(optimizer=reg.OPT_GN, max_iterations=4, max_correspondence_distance=2.0))
    ref_ = reg.BatchAPDGICP(reg.default_params(optimizer=reg.OPT_GN, max_iterations=4, max_correspondence_distance=2.0))
    ref_.set_clouds(0, [s, t])
    want_b = ref_.align([(0, 1)], [guess]).tobytes()
    tickets = []
    for rep_ in range(4):
        b.set_clouds(0, [torch.from_numpy(s).cuda() * 1.0, torch.from_numpy(t).cuda() * 1.0])
        junk = [torch.full(s.shape, 3e5, device="cuda") for _ in range(4)]
        tickets.append(b.align_enqueue([(0, 1)], [guess]))
        if len(tickets) == 2:
            assert b.align_collect(tickets.pop(0)).tobytes() == want_b
        del junk
    assert b.align_collect(tickets.pop(0)).tobytes() == want_b
    # many set_cloud calls with temporaries and no sync point in between: the handle lets a tensor go only after
    # waiting for the pack kernel that reads it
    big = torch.zeros(64 << 20, device="cuda")
    for rep_ in range(70):
        big.add_(1.0)
        b.set_cloud(0, torch.from_numpy(s).cuda() + big[: s.size].view(s.shape) * 0.0)
        junk = torch.full(s.shape, 7e5, device="cuda")
        del junk
    b.set_cloud(1, torch.from_numpy(t).cuda() * 1.0)
    assert len(b._keep_new) <= 65
    assert b.align([(0, 1)], [guess]).tobytes() == want_b
    del big


# ------------------------------------------------------------------ batched registrations (8e / C3)
def test_batch_matches_single(reg, golden, scene):
    clouds, pairs, guesses = [], [], []
    for i in range(5):
        n = (2048, 1500, 2300, 1024, 2048)[i]
        s, t, _, gs = scene.make_pair(n, n + 17 * i, scene.pair_seed(4, i), "odometry" if i % 2 == 0 else "loop")
        clouds += [s, t]
        pairs.append((2 * i, 2 * i + 1))
        guesses.append(gs)
    pairs.append((0, 3))          # clouds shared between pairs
    guesses.append(np.eye(4, dtype=np.float32))
    kw = dict(max_correspondence_distance=2.5, azimuth_variance_deg=1.0)
    b = reg.BatchAPDGICP(reg.default_params(**kw))
    for c in clouds:
        b.add_cloud(c)
    res = b.align(pairs, guesses)
    for i, (s, t) in enumerate(pairs):
        g = reg.FastAPDGICP(reg.default_params(**kw))
        g.setInputSource(clouds[s])
        g.setInputTarget(clouds[t])
        T = g.align(guesses[i])
        assert np.array_equal(reg.result_matrix(res[i]), T), i
        assert [res[i]["converged"], res[i]["iterations"], res[i]["n_linearize"], res[i]["n_compute_error"]] == info_of(g)
        o = R.RefAPDGICP(R.default_params(**kw))
        o.setInputSource(clouds[s])
        o.setInputTarget(clouds[t])
        To = o.align(guesses[i])
        te, re_ = scene.pose_error(To, T)
        assert te <= T_TOL and re_ <= R_TOL
    # one source against several keyframe targets (BASELINE configs[2])
    res2 = b.align([(0, 1), (0, 3), (0, 5), (0, 7)])
    assert res2["n_linearize"].min() >= 1


@pytest.mark.parametrize("mode", ("lm_launch", "gn6"))
def test_one_scan_against_eight_keyframes(reg, scene, mode):
    """BASELINE configs[2]: the newest scan registered against the last 8 keyframes of the same street in one batch; the
    scan is one cloud shared by all 8 pairs.  Every pair against the CPU oracle; batch == single-handle results."""
    src, tgts, _, guesses = scene.make_keyframe_set(1800, 1700, 8, 77)
    kw = (dict(max_correspondence_distance=2.0, transformation_epsilon=0.1, azimuth_variance_deg=1.0) if mode == "lm_launch" else
          dict(optimizer=1, max_iterations=6, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0))
    b = reg.BatchAPDGICP(reg.default_params(**kw))
    si = b.add_cloud(src)
    ti = [b.add_cloud(t) for t in tgts]
    res = b.align([(si, t) for t in ti], guesses)
    assert len(res) == 8
    for k in range(8):
        o = R.RefAPDGICP(R.default_params(**kw))
        o.setInputSource(src)
        o.setInputTarget(tgts[k])
        To = o.align(guesses[k])
        te, re_ = scene.pose_error(To, reg.result_matrix(res[k]))
        assert te <= T_TOL and re_ <= R_TOL, (k, te, re_)
        assert [res[k]["converged"], res[k]["iterations"], res[k]["n_linearize"]] == [int(o.converged), o.nr_iterations, o.n_linearize], k
    g = reg.FastAPDGICP(reg.default_params(**kw))
    g.setInputSource(src)
    for k in (0, 7):
        g.setInputTarget(tgts[k])
        assert np.array_equal(reg.result_matrix(res[k]), g.align(guesses[k])), k


@pytest.mark.parametrize("mode", ("lm_launch", "gn20"))
def test_c3_one_scan_against_eight_keyframes_at_full_size(reg, scene, mode):
    """BASELINE configs[2] at the size SURVEY 8d names: one 8192-point scan against the last 8 keyframes (8192 points each) in
    one batch, keyframe covariances cached, a new scan every call.  Flags and iteration counts exact against the oracle, poses
    inside north_star's tolerance (1e-3 m / 1e-4 rad), for the launch parameters (LM) and for 20 Gauss-Newton iterations."""
    src, tgts, _, guesses = scene.make_keyframe_set(8192, 8192, 8, scene.pair_seed(3, 0))
    kw = (dict(LAUNCH) if mode == "lm_launch" else
          dict(optimizer=1, max_iterations=20, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0, azimuth_variance_deg=1.0))
    b = reg.BatchAPDGICP(reg.default_params(**kw))
    si = b.add_cloud(src)
    ti = [b.add_cloud(t) for t in tgts]
    b.compute_covariances()
    pairs = b.make_pairs([(si, t) for t in ti], guesses)
    first = b.align(pairs).copy()
    b.set_cloud(si, src)          # the next frame's call: the scan is registered again, the keyframes stay
    res = b.align(pairs)
    assert res.tobytes() == first.tobytes()
    for k in range(8):
        o = R.RefAPDGICP(R.default_params(**kw))
        o.setInputSource(src)
        o.setInputTarget(tgts[k])
        To = o.align(guesses[k])
        te, re_ = scene.pose_error(To, reg.result_matrix(res[k]))
        assert te <= T_TOL and re_ <= R_TOL, (k, te, re_)
        assert [res[k]["converged"], res[k]["iterations"], res[k]["n_linearize"], res[k]["n_compute_error"]] == \
            [int(o.converged), o.nr_iterations, o.n_linearize, o.n_compute_error], k


@pytest.mark.parametrize("optimizer", ("gn", "lm"))
def test_two_batches_in_flight(reg, scene, optimizer):
    """apdgicp_batch_align_enqueue / _collect: batch s+1 (different clouds in the SAME slots) is set and enqueued before batch s is
    collected; every batch must equal its synchronous align bit for bit, in host and device form, and an error of one batch
    surfaces at its own collect."""
    import torch
    kw = (dict(optimizer=1, max_iterations=5, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0) if optimizer == "gn"
          else dict(max_correspondence_distance=2.0, transformation_epsilon=0.01))
    n_batches, n_pairs = 5, 6
    data = []
    for s in range(n_batches):
        clouds, guesses = [], []
        for p in range(n_pairs):
            a, b_, _, g = scene.make_pair(900 + 40 * p, 1000, scene.pair_seed(21 + s, p), "odometry")
            clouds += [a, b_]
            guesses.append(g)
        data.append((clouds, guesses))
    pair_idx = [(2 * i, 2 * i + 1) for i in range(n_pairs)]
    ref_b = reg.BatchAPDGICP(reg.default_params(**kw))
    want = []
    for clouds, guesses in data:
        ref_b.set_clouds(0, clouds)
        want.append(ref_b.align(pair_idx, guesses).copy())
    b = reg.BatchAPDGICP(reg.default_params(**kw))
    tickets = []
    for s, (clouds, guesses) in enumerate(data):
        b.set_clouds(0, clouds)
        tickets.append(b.align_enqueue(pair_idx, guesses))
        if s >= 1:  # collect the previous batch while this one runs
            got = b.align_collect(tickets[s - 1])
            assert got.tobytes() == want[s - 1].tobytes(), s - 1
            if s == 2:
                dev = b.align_collect(tickets[s - 1], device=True)  # collecting twice is allowed; device form
                assert dev.cpu().numpy().tobytes() == want[s - 1].tobytes()
    assert b.align_collect(tickets[-1]).tobytes() == want[-1].tobytes()
    pooled = optimizer == "lm" and os.environ.get("APDGICP_LM_POOL", "1") != "0" and os.environ.get("APDGICP_NN_MODE") != "brute"  # (tools/knob_matrix.sh)
    if not pooled:            # two record buffers: the ticket is void after the second enqueue behind its own
        with pytest.raises(Exception, match="ticket"):
            b.align_collect(tickets[0])
    else:                     # pooled LM batches: a lane per batch, eight of them
        assert b.align_collect(tickets[0]).tobytes() == want[0].tobytes()
    # a batch with non-finite points fails at ITS collect; the batches around it are unaffected
    bad = [c.copy() for c in data[0][0]]
    bad[3][5] = np.nan
    b.set_clouds(0, data[1][0])
    t_ok = b.align_enqueue(pair_idx, data[1][1])
    b.set_clouds(0, bad)
    with pytest.raises(Exception, match="non-finite"):
        # a run that polls as it goes (Gauss-Newton in several poll chunks, host-polled LM) fails in enqueue itself; a deferred
        # or pooled one hands the error out with its own batch, after the batch before it has been collected intact
        t_bad = b.align_enqueue(pair_idx, data[0][1])
        assert b.align_collect(t_ok).tobytes() == want[1].tobytes()
        b.align_collect(t_bad)
    assert b.align_collect(t_ok).tobytes() == want[1].tobytes()
    b.set_clouds(0, data[2][0])
    assert b.align(pair_idx, data[2][1]).tobytes() == want[2].tobytes()


def test_three_handles_keep_three_batches_in_flight(reg, scene):
    """bench.py's schedule: batch s on handle s % 3, one pair group per handle (apdgicp_batch_set_pair_groups), enqueued before
    the earlier ones are collected.  Every batch equals the synchronous align of a default handle bit for bit."""
    kw = dict(optimizer=1, max_iterations=6, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0)
    n_batches, n_pairs = 7, 24   # 24 pairs: a default handle splits them into three groups, these handles do not
    data = []
    for s in range(n_batches):
        clouds, guesses = [], []
        for p in range(n_pairs):
            a, b_, _, g = scene.make_pair(700 + 16 * (p % 5), 800, scene.pair_seed(31 + s % 3, p), "odometry")
            clouds += [a, b_]
            guesses.append(g)
        data.append((clouds, guesses))
    pair_idx = [(2 * i, 2 * i + 1) for i in range(n_pairs)]
    ref_b = reg.BatchAPDGICP(reg.default_params(**kw))
    want = []
    for clouds, guesses in data:
        ref_b.set_clouds(0, clouds)
        want.append(ref_b.align(pair_idx, guesses).copy())
    handles = [reg.BatchAPDGICP(reg.default_params(**kw)) for _ in range(3)]
    for h in handles:
        h.set_pair_groups(1)
    with pytest.raises(Exception):
        handles[0].set_pair_groups(0)
    tickets, got = [None] * 3, {}
    for s, (clouds, guesses) in enumerate(data):
        h = s % 3
        if tickets[h] is not None:
            got[tickets[h][0]] = handles[h].align_collect(tickets[h][1])
        handles[h].set_clouds(0, clouds)
        tickets[h] = (s, handles[h].align_enqueue(pair_idx, guesses))
    for h in range(3):
        if tickets[h] is not None:
            got[tickets[h][0]] = handles[h].align_collect(tickets[h][1])
    assert sorted(got) == list(range(n_batches))
    for s in range(n_batches):
        assert got[s].tobytes() == want[s].tobytes(), s


def test_large_batch_300_pairs(reg, scene):
    """More pairs than ride home with the status poll (256): the records come from the device buffer instead; the three
    pair groups, their separate covariance launches and a cloud shared by every pair are all in play."""
    n_pairs = 300
    tgt_shared = scene.make_pair(700, 900, scene.pair_seed(9, 0), "odometry")[1]
    clouds, pairs, guesses = [tgt_shared], [], []
    for i in range(n_pairs):
        s, t, _, gs = scene.make_pair(600 + (i % 7) * 31, 640, scene.pair_seed(9, 1 + i % 12), "odometry")
        clouds += [s, t]
        pairs.append((1 + 2 * i, 2 + 2 * i) if i % 10 else (1 + 2 * i, 0))
        guesses.append(gs)
    kw = dict(optimizer=1, max_iterations=4, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0)
    b = reg.BatchAPDGICP(reg.default_params(**kw))
    for c in clouds:
        b.add_cloud(c)
    res = b.align(pairs, guesses)
    assert len(res) == n_pairs and int(res["n_linearize"].min()) == 4
    for i in (0, 1, 10, 137, 299):
        g = reg.FastAPDGICP(reg.default_params(**kw))
        g.setInputSource(clouds[pairs[i][0]])
        g.setInputTarget(clouds[pairs[i][1]])
        assert np.array_equal(reg.result_matrix(res[i]), g.align(guesses[i])), i


def test_batch_is_deterministic(reg, scene):
    s, t, _, gs = scene.make_pair(2048, 2048, scene.pair_seed(4, 77), "odometry")
    b = reg.BatchAPDGICP(reg.default_params(**LAUNCH))
    i0, i1 = b.add_cloud(s), b.add_cloud(t)
    r1 = b.align([(i0, i1)] * 7, [gs] * 7)
    r2 = b.align([(i0, i1)] * 7, [gs] * 7)
    assert np.array_equal(r1["T"], r2["T"]) and np.all(r1["T"] == r1["T"][0])


# ------------------------------------------------------------------ exact pruning == brute force
def _fresh(reg, mode, **kw):
    """a handle whose engine was created with the given NN/kNN mode (read from the environment at creation)"""
    import os
    old = {k: os.environ.get(k) for k in ("APDGICP_NN_MODE", "APDGICP_KNN_MODE")}
    os.environ["APDGICP_NN_MODE"] = os.environ["APDGICP_KNN_MODE"] = mode
    try:
        return reg.FastAPDGICP(reg.default_params(**kw))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("sfx,flags", XF)
def test_pruned_search_is_bitwise_the_brute_force_search(reg, scene, sfx, flags):
    """The Z-curve/bounding-box pruning only skips work: covariances, correspondences, distances, H, b and
    the final pose must be IDENTICAL to the LDS-tiled brute-force kernels (north_star's formulation) -- under both
    summation orders of T * p."""
    for idx, (n, m, kind) in enumerate(((2048, 2048, "odometry"), (3000, 5000, "loop"), (8192, 8192, "odometry"))):
        src, tgt, _, guess = scene.make_pair(n, m, scene.pair_seed(6, idx), kind)
        kw = dict(max_correspondence_distance=2.5, azimuth_variance_deg=1.0, flags=flags)
        a, b = _fresh(reg, "pruned", **kw), _fresh(reg, "brute", **kw)
        for h in (a, b):
            h.setInputSource(src)
            h.setInputTarget(tgt)
        assert np.array_equal(a.getSourceCovariances(), b.getSourceCovariances())
        assert np.array_equal(a.getTargetCovariances(), b.getTargetCovariances())
        for T in (np.eye(4), guess.astype(np.float64)):
            ra, rb = a.linearize(T), b.linearize(T)
            ca, sa = a.correspondences()
            cb, sb = b.correspondences()
            assert np.array_equal(ca, cb) and np.array_equal(sa.view(np.uint32), sb.view(np.uint32))
            assert ra[0] == rb[0] and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2])
        Ta, Tb = a.align(guess), b.align(guess)
        assert np.array_equal(Ta, Tb) and info_of(a) == info_of(b)


def _handle_with_env(reg, cls, env, **kw):
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return cls(reg.default_params(**kw))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.gpu
@pytest.mark.parametrize("kind,flags", (("odometry", 0), ("loop", 0), ("loop", 2)))
def test_kept_neighbours_are_the_searched_neighbours(reg, scene, kind, flags):
    """Neighbour keeping (nn_search: a point whose previous neighbour is PROVEN to still be the nearest skips the search)
    must not change a single bit: GN-20 and LM runs with the skin on (several settings), off, and with the brute-force
    search; the counters must show that points really were kept."""
    gn = dict(optimizer=1, max_iterations=20, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0,
              azimuth_variance_deg=1.0, flags=flags)
    lm = dict(max_correspondence_distance=2.0, transformation_epsilon=1e-4, azimuth_variance_deg=1.0, flags=flags)
    clouds, pairs, guesses = [], [], []
    for i in range(6):
        n = (8192, 4100, 2048, 3000, 8192, 1500)[i]
        s_, t_, _, g_ = scene.make_pair(n, n + 100 * i, scene.pair_seed(21, i), kind)
        clouds += [s_, t_]
        pairs.append((2 * i, 2 * i + 1))
        guesses.append(g_)
    envs = ({"APDGICP_NN_SKIN": "0"}, {}, {"APDGICP_NN_W": "2"}, {"APDGICP_NN_W": "4"}, {"APDGICP_NN_W": "8"}, {"APDGICP_NN_MODE": "brute"},
            # one pair group per handle = the throughput regime of bench.py: k_nn_compact (blocks of 256 points that pack the
            # points still searching into fewer waves; a block with one wave's worth of points left searches them with all
            # four waves) -- APDGICP_NN_W=1: six pairs are too few for the engine to choose that regime by itself -- and the
            # same regime without keeping (one-wave blocks of k_nn_pruned)
            {"ONE_GROUP": "1", "APDGICP_NN_W": "1"}, {"ONE_GROUP": "1", "APDGICP_NN_W": "1", "APDGICP_NN_SKIN": "0"}, {"APDGICP_NN_W": "1"},
            # ... and its blocks with only a FEW points left: the point-serial path (targets of at most 8192 points; default: up to 32
            # points per block) off, for every block the cooperative search would take, and for a handful of points only
            {"ONE_GROUP": "1", "APDGICP_NN_W": "1", "APDGICP_NN_SPARSE": "0"}, {"ONE_GROUP": "1", "APDGICP_NN_W": "1", "APDGICP_NN_SPARSE": "64"},
            {"ONE_GROUP": "1", "APDGICP_NN_W": "1", "APDGICP_NN_SPARSE": "5"})
    for kw in (gn, lm):
        want = None
        chunk_scans = {}
        for env in envs:
            env = dict(env)
            one_group = env.pop("ONE_GROUP", None)
            b = _handle_with_env(reg, reg.BatchAPDGICP, dict(env, APDGICP_STATS="1"), **kw)
            if one_group:
                b.set_pair_groups(1)
            b.set_clouds(0, clouds)
            res = b.align(pairs, guesses)
            st = b.debug_stats()
            got = res.tobytes()
            want = want or got
            assert got == want, (env, kw is gn)
            if "APDGICP_NN_SPARSE" in env:
                chunk_scans[env["APDGICP_NN_SPARSE"]] = st[2]
        if not any(os.environ.get(v) for v in ("APDGICP_NN_SKIN", "APDGICP_NN_MODE", "APDGICP_NN_SPARSE")):
            # the point-serial path scans no chunks: the more blocks take it, the fewer chunk scans the other paths are left with
            assert chunk_scans["64"] < chunk_scans["5"] < chunk_scans["0"], chunk_scans
            if env.get("APDGICP_NN_SKIN") == "0" or env.get("APDGICP_NN_MODE") == "brute":
                assert st[6] == 0
            elif kw is gn and not any(os.environ.get(v) for v in ("APDGICP_NN_SKIN", "APDGICP_NN_MODE")):   # (tools/knob_matrix.sh switches that turn keeping off)
                assert st[6] > 0.3 * 18 * sum(len(clouds[2 * i]) for i in range(6)), st   # most points, most iterations
    # large targets (> 16384 points: the super-box level) through the one-group path: k_nn_compact's waves walk the batches of
    # group boxes on their own, without block barriers -- same records as the multi-wave k_nn_pruned blocks and as no keeping
    s_, t_, _, g_ = scene.make_pair(6000, 40_000, scene.pair_seed(21, 60), kind)
    s2, t2, _, g2 = scene.make_pair(20_000, 30_000, scene.pair_seed(21, 61), kind)
    want = None
    for env, one_group in (({}, False), ({"APDGICP_NN_W": "1"}, True), ({"APDGICP_NN_W": "1", "APDGICP_NN_SKIN": "0"}, True)):
        b = _handle_with_env(reg, reg.BatchAPDGICP, env, **dict(gn, max_iterations=8))
        if one_group:
            b.set_pair_groups(1)
        b.set_clouds(0, [s_, t_, s2, t2])
        got = b.align([(0, 1), (2, 3)], [g_, g2]).tobytes()
        want = want or got
        assert got == want, (env, one_group)
    # a single handle: correspondences, distances and H, b after the last linearize of an align are those of a cold search
    s_, t_, _, g_ = scene.make_pair(8192, 8192, scene.pair_seed(21, 50), kind)
    a = _handle_with_env(reg, reg.FastAPDGICP, {}, **gn)
    z = _handle_with_env(reg, reg.FastAPDGICP, {"APDGICP_NN_SKIN": "0"}, **gn)
    out = []
    for h in (a, z):
        h.setInputSource(s_), h.setInputTarget(t_)
        T = h.align(g_)
        c, q = h.correspondences()
        out.append((T, c, q.view(np.uint32), h.mahalanobis(), h.getFinalHessian()))
    for x, y in zip(*out):
        assert np.array_equal(x, y)


def test_tiled_sort_from_host_and_device_clouds(reg, scene):
    """The tiled register sort (k_sort_tiles / k_merge_tiles / k_boxes_sorted, 2048 < n <= 16384) reads a host cloud straight
    from pinned memory with the bounding box reduced on the host, a device cloud from the pack kernel's output with the box
    reduced on the device: both must give the permutation the brute-force kernels see -- everything downstream
    (covariances, correspondences, fp32 distances, H, b) is bitwise the same, in all three tile classes and at their edges."""
    import torch
    for n, m in ((2049, 4096), (4097, 8192), (8193, 12000), (16384, 9000)):
        src, tgt, _, guess = scene.make_pair(n, m, scene.pair_seed(8, n), "odometry")
        out = []
        for form, env in (("host", {}), ("device", {}), ("host", {"APDGICP_NN_MODE": "brute", "APDGICP_KNN_MODE": "brute"})):
            g = _handle_with_env(reg, reg.FastAPDGICP, env, max_correspondence_distance=2.0)
            g.setInputSource(src if form == "host" else torch.from_numpy(src).cuda())
            g.setInputTarget(tgt if form == "host" else torch.from_numpy(tgt).cuda())
            c, H, b = g.linearize(guess.astype(np.float64))
            corr, sqd = g.correspondences()
            out.append((g.getSourceCovariances(), g.getTargetCovariances(), corr, sqd, c, H, b))
        for other in out[1:]:
            for x, y in zip(out[0], other):
                assert np.array_equal(np.asarray(x), np.asarray(y))


def test_pinned_cloud_buffers_are_reused_safely(reg, scene):
    """Scan-sized host clouds stay in the slot's pinned buffer until the sort has read them (TileJob::staged): replacing a
    cloud before any align, after an align, and with another size class must always register the LATEST points."""
    kw = dict(max_correspondence_distance=2.0, azimuth_variance_deg=1.0)
    sets = [scene.make_pair(n, m, scene.pair_seed(9, n), "odometry") for n, m in ((8192, 8192), (5000, 3000), (8192, 12000), (1500, 9000))]

    def fresh(src, tgt, guess):
        g = reg.FastAPDGICP(reg.default_params(**kw))
        g.setInputSource(src), g.setInputTarget(tgt)
        return g.align(guess), g.getSourceCovariances()

    h = reg.FastAPDGICP(reg.default_params(**kw))
    for rnd in range(2):
        for k, (src, tgt, _, guess) in enumerate(sets):
            other = sets[(k + 1) % len(sets)]
            h.setInputSource(other[0])          # replaced before it is ever used
            h.setInputSource(src)
            h.setInputTarget(tgt)
            T = h.align(guess)
            Tw, cw = fresh(src, tgt, guess)
            assert np.array_equal(T, Tw), (rnd, k)
            assert np.array_equal(h.getSourceCovariances(), cw), (rnd, k)
            # same target, new scan (the odometry loop): only the source buffer is rewritten
            src2 = (src + np.float32(0.01)).astype(np.float32)
            h.setInputSource(src2)
            T2 = h.align(guess)
            assert np.array_equal(T2, fresh(src2, tgt, guess)[0]), (rnd, k)


def test_a_host_cloud_replaced_by_a_device_cloud_before_any_align(reg, scene):
    """A scan-sized host cloud stays in the slot's pinned buffer until the sort reads it (`staged`); replacing the slot through
    apdgicp_batch_set_clouds(on_device = 1) -- or clearing it -- before any align must void that pinned copy: the registration has
    to use the LATEST points (the flag used to survive, and the sort read the old host cloud)."""
    import torch
    kw = dict(max_correspondence_distance=2.0, azimuth_variance_deg=1.0)
    old_src, old_tgt, _, _ = scene.make_pair(8192, 8192, scene.pair_seed(13, 0), "odometry")
    src, tgt, _, guess = scene.make_pair(8192, 6000, scene.pair_seed(13, 1), "odometry")
    b = reg.BatchAPDGICP(reg.default_params(**kw))
    b.set_cloud(0, old_src)          # host clouds of the tiled-sort size class: staged, not yet sorted
    b.set_cloud(1, old_tgt)
    b.set_clouds(0, [torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()])   # same slots, device memory, another size
    got = b.align([(0, 1)], [guess])
    fresh = reg.BatchAPDGICP(reg.default_params(**kw))
    fresh.set_clouds(0, [src, tgt])
    want = fresh.align([(0, 1)], [guess])
    assert got.tobytes() == want.tobytes()
    # and the other way round: device first, then a host cloud into the same slot
    b.set_clouds(0, [torch.from_numpy(old_src).cuda(), torch.from_numpy(old_tgt).cuda()])
    b.set_cloud(0, src), b.set_cloud(1, tgt)
    assert b.align([(0, 1)], [guess]).tobytes() == want.tobytes()


def test_many_host_clouds_packed_by_the_host_pool(reg, scene):
    """apdgicp_batch_set_clouds with HOST clouds packs them on a few host threads (four or more scan-sized clouds): records byte for
    byte those of the same clouds set one by one (no pool) and of device-resident copies -- for tight [n, 3] rows, padded rows
    (pcl::PointXYZI: 32 bytes), mixed sizes including one outside the pinned size class, repeated on the same slots."""
    import torch
    kw = dict(optimizer=1, max_iterations=4, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0)
    sizes = [(8192, 8192), (4096, 8192), (8192, 2049), (3000, 8192), (8192, 8192), (1500, 20000)]
    clouds, guesses = [], []
    for i, (n, m) in enumerate(sizes):
        s_, t_, _, g_ = scene.make_pair(n, m, scene.pair_seed(21, i), "odometry")
        clouds += [s_, t_]
        guesses.append(g_)
    pairs = [(2 * i, 2 * i + 1) for i in range(len(sizes))]

    def padded(c):
        buf = np.full((len(c), 8), 7.0, dtype=np.float32)   # x y z pad intensity pad pad pad
        buf[:, :3] = c
        return buf

    one = reg.BatchAPDGICP(reg.default_params(**kw))
    for i, c in enumerate(clouds):
        one.set_cloud(i, c)
    want = one.align(pairs, guesses).tobytes()
    dev = reg.BatchAPDGICP(reg.default_params(**kw))
    dev.set_clouds(0, [torch.from_numpy(c).cuda() for c in clouds])
    assert dev.align(pairs, guesses).tobytes() == want
    many = reg.BatchAPDGICP(reg.default_params(**kw))
    for rnd in range(3):
        many.set_clouds(0, [padded(c) for c in clouds] if rnd == 1 else clouds)
        assert many.align(pairs, guesses).tobytes() == want, rnd
    # the same slots with OTHER clouds in between: the pinned buffers are rewritten, nothing stale survives
    many.set_clouds(0, clouds[::-1])
    many.align([(1, 0)], [guesses[-1]])
    many.set_clouds(0, clouds)
    assert many.align(pairs, guesses).tobytes() == want


def test_exact_ties_resolve_to_the_lowest_original_index(reg):
    """A source point exactly midway between two target points (equal fp32 distances, far apart on the
    Z-curve) and duplicated target points: the oracle's rule is (distance, index) lexicographic."""
    rng = np.random.default_rng(11)
    filler = rng.uniform(-40, 40, size=(3000, 3)).astype(np.float32) + np.float32(100.0)
    pairs_l = np.stack([np.full(64, -1.0), np.arange(64) * 2.0, np.zeros(64)], axis=1).astype(np.float32)
    pairs_r = pairs_l * np.array([-1, 1, 1], dtype=np.float32)
    tgt = np.concatenate([filler[:1500], pairs_r, filler[1500:], pairs_l, pairs_r[:16]])  # right copies come first; 16 exact duplicates last
    src = np.concatenate([np.stack([np.zeros(64), np.arange(64) * 2.0, np.zeros(64)], axis=1).astype(np.float32), filler[:500] + np.float32(0.25)])
    o = R.RefAPDGICP(R.default_params(max_correspondence_distance=3.0))
    o.setInputSource(src)
    o.setInputTarget(tgt)
    o.linearize(np.eye(4))
    co, so = o.correspondences()
    assert np.array_equal(co[:64], 1500 + np.arange(64))      # lowest index among the tied targets
    for mode in ("pruned", "brute"):
        g = _fresh(reg, mode, max_correspondence_distance=3.0)
        g.setInputSource(src)
        g.setInputTarget(tgt)
        g.linearize(np.eye(4))
        cg, sg = g.correspondences()
        assert np.array_equal(cg, co), mode
        assert np.array_equal(sg.view(np.uint32), so.view(np.uint32)), mode
        assert np.abs(g.getTargetCovariances()[:, :3, :3] - o.covariances("target")).max() <= 1e-10, mode
    # every shape of the search kernel settles the ties itself (nn_search: the wave looks through the target together), and
    # identically: the batch records of a short Gauss-Newton run -- its first tick sees all 64 ties -- are byte-equal to those
    # of the brute-force kernels, whose ties k_linearize resolves
    gn = dict(optimizer=1, max_iterations=3, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=3.0)
    want = None
    for env, one_group in (({"APDGICP_NN_MODE": "brute"}, False), ({}, False), ({"APDGICP_NN_W": "1"}, True), ({"APDGICP_NN_W": "2"}, False),
                           ({"APDGICP_NN_W": "4"}, False), ({"APDGICP_NN_W": "8"}, False), ({"APDGICP_NN_W": "1", "APDGICP_NN_SKIN": "0"}, True),
                           ({"APDGICP_NN_W": "1", "APDGICP_NN_SPARSE": "0"}, True), ({"APDGICP_NN_W": "1", "APDGICP_NN_SPARSE": "64"}, True)):
        b = _handle_with_env(reg, reg.BatchAPDGICP, env, **gn)
        if one_group:
            b.set_pair_groups(1)
        b.set_clouds(0, [src, tgt, src, tgt])
        got = b.align([(0, 1), (2, 3)]).tobytes()
        want = want or got
        assert got == want, env


def test_large_cloud_generic_sort_path(reg, scene):
    """> 16384 points: the clouds are sorted by the global-memory bitonic path instead of the LDS one."""
    src, tgt, _, guess = scene.make_pair(20000, 24000, scene.pair_seed(6, 9), "odometry")
    kw = dict(max_correspondence_distance=2.0, azimuth_variance_deg=1.0, max_iterations=3)
    g, o = both(reg, src, tgt, **kw)
    c1, H1, b1 = g.linearize(guess.astype(np.float64))
    c2, H2, b2 = o.linearize(guess.astype(np.float64))
    cg, sg = g.correspondences()
    co, so = o.correspondences()
    assert np.array_equal(cg, co) and np.array_equal(sg.view(np.uint32), so.view(np.uint32))
    assert rel_err(H1, H2) < HB_TOL and rel_err(b1, b2) < HB_TOL
    assert np.abs(g.getSourceCovariances()[:, :3, :3] - o.covariances("source")).max() <= 1e-10


def test_c5_dense_submap_gn20(reg, scene):
    """BASELINE configs[4] as SURVEY 8d states it: 100k-point source against a 500k-point accumulated map (generic sort path,
    super boxes, many group-box batches), 20 Gauss-Newton iterations, against the CPU oracle: correspondences and fp32
    distances at the guess bit-exact, H / b / cost HB_TOL, exactly 20 linearisations, final pose inside the north-star tolerance."""
    src, tgt, _, guess = scene.make_pair(100_000, 500_000, scene.pair_seed(5, 0), "odometry")
    kw = dict(optimizer=1, max_iterations=20, transformation_epsilon=1e-300, rotation_epsilon=1e-300,
              max_correspondence_distance=2.0, azimuth_variance_deg=1.0)
    g, o = both(reg, src, tgt, **kw)
    c1, H1, b1 = g.linearize(guess.astype(np.float64))
    c2, H2, b2 = o.linearize(guess.astype(np.float64))
    cg, sg = g.correspondences()
    co, so = o.correspondences()
    assert np.array_equal(sg.view(np.uint32), so.view(np.uint32))
    assert np.array_equal(cg, co)
    assert rel_err(H1, H2) < HB_TOL and rel_err(b1, b2) < HB_TOL and abs(c1 - c2) < HB_TOL * c2
    T, To = g.align(guess), o.align(guess)
    assert g.result.n_linearize == 20 == o.n_linearize and g.result.iterations == 19 == o.nr_iterations
    te, re_ = scene.pose_error(To, T)
    print("C5 GN-20 pose difference vs oracle", te, re_)
    assert te <= T_TOL and re_ <= R_TOL
    cg, sg = g.correspondences()      # after the 20th linearize: the same correspondences as the oracle's last update
    co, so = o.correspondences()
    assert np.mean(cg == co) > 0.9999


def test_dense_search_block_order_changes_nothing(reg, scene, monkeypatch):
    """Round 6: from the third tick of a dense one-pair registration on, the search blocks are launched costliest first (k_block_order; more
    than 1280 blocks = more than 81 920 source points against a target beyond the scan class).  Which block runs when never changes a record:
    GN-6 and LM with the order on (default) and off (APDGICP_NN_ORDER=0) -- results byte for byte, correspondences and distances too; and the
    order really is a permutation of the blocks, heaviest first (read back through the C ABI's debug hook)."""
    src, tgt, _, guess = scene.make_pair(90_000, 120_000, scene.pair_seed(5, 1), "odometry")
    for kw in (dict(optimizer=1, max_iterations=6, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0, azimuth_variance_deg=1.0), LAUNCH):
        out = []
        for order in ("1", "0"):
            monkeypatch.setenv("APDGICP_NN_ORDER", order)
            b = reg.BatchAPDGICP(reg.default_params(**kw))
            b.set_clouds(0, [src, tgt])
            r = b.align([(0, 1)], [guess])
            out.append(r.tobytes())
            del b
        assert out[0] == out[1]
    monkeypatch.delenv("APDGICP_NN_ORDER")


# ------------------------------------------------------------------ SURVEY 8f rows
def test_plain_gicp_mode(reg, golden, scene):
    """f4: APDGICP_FLAG_PLAIN_GICP == upstream fast_gicp::FastGICP (cov_dist = 0); without the fp32 atan2f the
    smooth outputs agree much tighter."""
    src, tgt, guess = golden["lin_source"], golden["lin_target"], golden["lin_guess"]
    kw = dict(max_correspondence_distance=2.5, flags=reg.FLAG_PLAIN_GICP)
    g, o = both(reg, src, tgt, **kw)
    c1, H1, b1 = g.linearize(guess.astype(np.float64))
    c2, H2, b2 = o.linearize(guess.astype(np.float64))
    assert np.array_equal(g.correspondences()[0], o.correspondences()[0])
    assert rel_err(H1, H2) < 1e-10 and rel_err(b1, b2) < 1e-9 and abs(c1 - c2) < 1e-10 * c2
    T, To = g.align(guess), o.align(guess)
    assert info_of(g) == [int(o.converged), o.nr_iterations, o.n_linearize, o.n_compute_error]
    te, re_ = scene.pose_error(To, T)
    assert te <= T_TOL and re_ <= R_TOL


def test_batch_fitness_and_candidate_verifier(reg, scene, pkg):
    """f1/f2: batched getFitnessScore and the LoopDetector::matching selection rule (loop_detector.cpp:404-431)."""
    import apdgicp_np as O
    lv = importlib.import_module("riv-slam_amd.loop_verifier")
    cands, guesses = [], []
    tgt = None
    for i in range(4):
        s, t, _, g = scene.make_pair(2048, 2048, scene.pair_seed(9, 0 if i != 2 else 50), "odometry")
        if tgt is None:
            tgt = t
        # candidate 2 comes from another scene (must lose), candidates 1 and 3 are noisy copies of candidate 0
        noise = (0.0, 0.02, 0.0, 0.05)[i]
        cands.append(s + np.float32(noise) * np.random.default_rng(3 + i).standard_normal(s.shape).astype(np.float32))
        guesses.append(g)
    kw = dict(max_correspondence_distance=2.5, azimuth_variance_deg=1.0)
    b = reg.BatchAPDGICP(reg.default_params(**kw))
    loop, scores, results = lv.verify_candidates(b, tgt, cands, guesses, fitness_score_max_range=4.0, fitness_score_thresh=10.0)
    # scores against an independent evaluation of pcl's definition
    for i, c in enumerate(cands):
        T = reg.result_matrix(results[i])
        pt = O.transform_points_f32(T.astype(np.float64), c)
        _, sq = O.nn1(pt, tgt)
        sel = sq.astype(np.float64) <= 4.0
        assert abs(scores[i] - sq[sel].astype(np.float64).mean()) < 1e-9 * scores[i]
    conv = [bool(r["converged"]) for r in results]
    want = -1
    for i in range(4):   # loop_detector.cpp:416: `if(!converged || score > best_score) continue;` -> the later of equal scores wins
        if conv[i] and (want < 0 or scores[i] <= scores[want]):
            want = i
    assert loop is not None and loop.candidate == want and loop.candidate != 2
    assert np.array_equal(loop.relative_pose, reg.result_matrix(results[want]))
    none, _, _ = lv.verify_candidates(b, tgt, cands, guesses, fitness_score_max_range=4.0, fitness_score_thresh=1e-9)
    assert none is None
    # explicit poses: identity
    s2, _ = b.fitness([(1, 0)], np.eye(4)[None], 4.0)
    _, sq = O.nn1(cands[0], tgt)
    sel = sq.astype(np.float64) <= 4.0
    assert abs(s2[0] - sq[sel].astype(np.float64).mean()) < 1e-9 * s2[0]


def test_c99_program_registers_through_the_abi():
    """tests/c/abi_smoke.c (plain C, gcc -std=c99) runs a registration through the single-handle ABI on the GPU"""
    import subprocess
    from test_abi_cpu import _build_c_smoke
    exe = _build_c_smoke()
    out = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "converged 1" in out.stdout


def test_measurement_hooks(reg, scene):
    """the hooks bench.py's roofline leg relies on: sampled kernel-timestamp timing of the search launches, tick count,
    pruning counters (APDGICP_STATS); profiling must not change the results"""
    import os
    clouds, pairs, guesses = [], [], []
    for i in range(8):
        s, t, _, gs = scene.make_pair(2048, 2048, scene.pair_seed(12, i), "odometry")
        clouds += [s, t]
        pairs.append((2 * i, 2 * i + 1))
        guesses.append(gs)
    kw = dict(optimizer=1, max_iterations=10, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0)
    os.environ["APDGICP_STATS"] = "1"
    try:
        b = reg.BatchAPDGICP(reg.default_params(**kw))
    finally:
        os.environ.pop("APDGICP_STATS", None)
    for c in clouds:
        b.add_cloud(c)
    ref = b.align(pairs, guesses)
    b.set_profiling(True)
    res = b.align(pairs, guesses)
    assert np.array_equal(np.asarray(res["T"]), np.asarray(ref["T"]))
    ms, launches, pair_iters = b.last_nn_profile()
    ticks, s_per_lane, splits = b.last_ticks()
    default_shape = not os.environ.get("APDGICP_NN_MODE")  # tools/knob_matrix.sh
    assert ticks == 10 and (not default_shape or (s_per_lane == 1 and splits == 1))
    assert launches >= 1 and 0.0 < ms / launches < 5.0          # a search launch takes tens of microseconds
    assert 1 <= pair_iters <= 8 * 10
    if os.environ.get("APDGICP_NN_MODE") != "brute":            # (the brute-force kernels have nothing to count)
        st = b.debug_stats()
        assert st[3] > 0 and st[2] > 0 and st[2] <= st[1]       # waves, chunks scanned <= chunks tested
        assert b.debug_stats()[3] == 0                          # reading resets


@pytest.mark.parametrize("n", (8192, 100_000))
def test_self_registration_properties_at_full_size(reg, scene, n):
    """Size-independent properties at BASELINE's sizes (no oracle needed): a cloud registered against itself keeps every
    point's own index at distance exactly 0, costs exactly 0 and returns exactly the identity; against a rigidly moved
    copy of itself the true motion comes back and every correspondence is the point itself."""
    cloud, _, _, _ = scene.make_pair(n, 16, scene.pair_seed(13, n), "odometry")
    kw = dict(max_correspondence_distance=2.0, azimuth_variance_deg=1.0)
    g = reg.FastAPDGICP(reg.default_params(**kw))
    g.setInputSource(cloud)
    g.setInputTarget(cloud.copy())
    cost, H, b = g.linearize(np.eye(4))
    corr, sqd = g.correspondences()
    uniq = np.unique(cloud, axis=0).shape[0] == n
    assert not sqd.any() and cost == 0.0 and not b.any()
    if uniq:
        assert np.array_equal(corr, np.arange(n))
    T = g.align(None)
    assert np.array_equal(T, np.eye(4, dtype=np.float32)) and g.hasConverged() and g.result.final_cost == 0.0
    # rigidly moved copy: target = R * cloud + t  ->  align(source = cloud) returns (R, t)
    M = scene.make_transform(np.array([0.3, -0.1, 0.02]), np.deg2rad(1.0), np.deg2rad(0.2), np.deg2rad(-0.1))
    moved = ((M[:3, :3] @ cloud[:, :3].astype(np.float64).T).T + M[:3, 3]).astype(np.float32)
    h = reg.FastAPDGICP(reg.default_params(**kw))
    h.setInputSource(cloud)
    h.setInputTarget(moved)
    T2 = h.align(None)
    te, re_ = scene.pose_error(M, T2)
    # against the TRUE motion the bar is the optimiser's own stopping tolerance (rotation_epsilon 2e-3, translation 5e-4)
    # plus the fp32 rounding of the moved copy, not the parity bar
    assert te <= 1e-3 and re_ <= 5e-4 and h.hasConverged(), (te, re_)
    corr2, _ = h.correspondences()
    assert (corr2 == np.arange(n)).mean() > 0.999


def test_covariances_are_equivariant_at_8k(reg, scene):
    """a6 at BASELINE's size without the oracle: cov(R p + t) = R cov(p) R^T for the unregularised neighbourhood covariance
    (the moved cloud is rounded to fp32, which can swap near-tied k-th neighbours: asserted for 99 % of the points)"""
    cloud, _, _, _ = scene.make_pair(8192, 16, scene.pair_seed(14, 0), "odometry")
    M = scene.make_transform(np.array([1.0, -2.0, 0.3]), np.deg2rad(25.0), np.deg2rad(3.0), np.deg2rad(-2.0))
    R3 = M[:3, :3]
    moved = ((R3 @ cloud[:, :3].astype(np.float64).T).T + M[:3, 3]).astype(np.float32)
    covs = []
    for c in (cloud, moved):
        g = reg.FastAPDGICP(reg.default_params(regularization=0))
        g.setInputSource(c)
        covs.append(g.getSourceCovariances()[:, :3, :3])
    want = np.einsum("ij,njk,lk->nil", R3, covs[0], R3)
    err = np.abs(covs[1] - want).reshape(len(cloud), -1).max(1) / np.maximum(np.abs(want).reshape(len(cloud), -1).max(1), 1e-12)
    assert (err < 1e-3).mean() > 0.99, float((err < 1e-3).mean())



def test_adversarial_fuzz_runs_clean_for_a_few_seconds():
    """tests/measure/fuzz_parity.py (lattices, duplicates, lines, planes, 3 km offsets, block-edge sizes, tiny clouds; every
    regularisation, both transform orders, GN and LM) against the oracle: about 300 cases here, 5 000 per run in profiles/."""
    import json
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "measure", "fuzz_parity.py")
    out = subprocess.run([sys.executable, tool, "20", "100000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout)
    assert d["n_failures"] == 0 and sum(k["cases"] for k in d["kinds"].values()) >= 50


# ------------------------------------------------------------------ round 5: batched nearest neighbours, the fp32 per-point mode
@pytest.mark.parametrize("sfx,flags", XF)
def test_nearest_neighbours_are_the_brute_force_ones(reg, golden, sfx, flags):
    """apdgicp_nearest_neighbours (what serves pcl::search::KdTree::nearestKSearch(pt, 1, ...) for the base-class calls of the
    nodelets): index and fp32 squared distance of the nearest target point of every T-transformed source point, no gate --
    against the numpy restatement's brute-force search (FLANN L2_Simple order, lowest index on ties), bit for bit."""
    import apdgicp_np as O
    src, tgt = golden["lin_source"], golden["lin_target"]
    g = reg.FastAPDGICP(reg.default_params(flags=flags, max_correspondence_distance=0.5))   # (a tight gate must not matter)
    g.setInputSource(src)
    g.setInputTarget(tgt)
    for T in (np.eye(4), golden["lin_guess"].astype(np.float64), golden["lin_T_true"]):
        idx, sqd = g.nearestNeighbours(T.astype(np.float32))
        pt = O.transform_points_f32(T.astype(np.float32).astype(np.float64), src, bool(flags & 2))
        want_i, want_d = O.nn1(pt, tgt)
        assert np.array_equal(idx, want_i) and np.array_equal(sqd.view(np.uint32), want_d.view(np.uint32))
    assert np.array_equal(g.getPoints(reg.TARGET), tgt) and np.array_equal(g.getPoints(reg.SOURCE), src)
    big_s, big_t, _, guess = __import__("importlib").import_module("riv-slam_amd.scene").make_pair(9000, 20000, 77, "odometry")
    g.setInputSource(big_s)
    g.setInputTarget(big_t)                                    # (scan-sized staged source, generic-sort target)
    idx, sqd = g.nearestNeighbours(guess)
    pt = O.transform_points_f32(guess.astype(np.float64), big_s, bool(flags & 2))
    want_i, want_d = O.nn1(pt, big_t)
    assert np.array_equal(idx, want_i) and np.array_equal(sqd.view(np.uint32), want_d.view(np.uint32))
    assert np.array_equal(g.getPoints(reg.TARGET), big_t)


def test_nearest_neighbours_of_arbitrary_queries_are_the_brute_force_ones(reg, golden):
    """apdgicp_nearest_neighbours_of (round 6: what serves pcl::search::Search::nearestKSearch(cloud, indices, 1, ...) and a foreign
    query point): queries that are NOT the source -- inside the target's box, far outside it, on target points (distance 0), duplicated,
    one single query, more queries than the target has points, rows of 12 and of 16 bytes -- against the numpy restatement's brute-force
    search, index and fp32 squared distance bit for bit; the handle's own source / target and its next align are untouched."""
    import apdgicp_np as O
    src, tgt, guess = golden["lin_source"], golden["lin_target"], golden["lin_guess"]
    g = reg.FastAPDGICP(reg.default_params(**LAUNCH))
    g.setInputSource(src)
    g.setInputTarget(tgt)
    T0 = g.align(guess).copy()
    rng = np.random.default_rng(606)
    lo, hi = tgt.min(0), tgt.max(0)
    inside = (lo + (hi - lo) * rng.random((3000, 3))).astype(np.float32)
    far = (inside[:200] + np.float32(1000.0) * rng.choice([-1.0, 1.0], (200, 3))).astype(np.float32)
    on = tgt[rng.integers(0, len(tgt), 300)]
    q = np.concatenate([inside, far, on, inside[:50], inside[:50]]).astype(np.float32)
    for queries in (q, q[:1], np.concatenate([q] * (len(tgt) // len(q) + 2))):
        idx, sqd = g.nearestNeighboursOf(queries)
        want_i, want_d = O.nn1(queries, tgt)
        assert np.array_equal(sqd.view(np.uint32), want_d.view(np.uint32))
        assert np.array_equal(idx, want_i)
    assert (sqd[3200:3500] == 0).all()                           # (the `on` block of the long query set's first copy)
    q16 = np.zeros((len(q), 4), dtype=np.float32)
    q16[:, :3], q16[:, 3] = q, 7.0
    idx16, sqd16 = g.nearestNeighboursOf(q16)                  # (16-byte rows: the fourth float is not a coordinate)
    want_i, want_d = O.nn1(q, tgt)
    assert np.array_equal(idx16, want_i) and np.array_equal(sqd16.view(np.uint32), want_d.view(np.uint32))
    assert np.array_equal(g.getPoints(reg.TARGET), tgt) and np.array_equal(g.getPoints(reg.SOURCE), src)
    assert np.array_equal(g.align(guess), T0)                      # (the scratch cloud slot did not disturb the handle's pair)
    with pytest.raises(reg.ApdgicpError):
        g.nearestNeighboursOf(np.zeros((0, 3), dtype=np.float32))
    h = reg.FastAPDGICP(reg.default_params())
    with pytest.raises(reg.ApdgicpError):
        h.nearestNeighboursOf(q)                                  # no target


def test_fp32_point_math_is_an_opt_in_inside_the_tolerance(reg, golden, scene):
    """APDGICP_FLAG_FP32_POINT_MATH: same correspondences and distances at a pose (the search is untouched), H / b / cost within
    1e-4 of the default's (fp32 algebra behind the search), final poses within 1e-5 m / 1e-6 rad of the default's and inside
    the north-star tolerance against the oracle; the default is unchanged by the flag's existence (golden tests)."""
    src, tgt, guess = golden["lin_source"], golden["lin_target"], golden["lin_guess"]
    for kw in (LAUNCH, dict(optimizer=1, max_iterations=20, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0)):
        a = reg.FastAPDGICP(reg.default_params(**kw))
        b = reg.FastAPDGICP(reg.default_params(flags=reg.FLAG_FP32_POINT_MATH, **kw))
        o = R.RefAPDGICP(R.default_params(**kw))
        for x in (a, b, o):
            x.setInputSource(src)
            x.setInputTarget(tgt)
        T0 = guess.astype(np.float64)
        c1, H1, b1 = a.linearize(T0)
        c2, H2, b2 = b.linearize(T0)
        assert np.array_equal(a.correspondences()[0], b.correspondences()[0])
        assert np.array_equal(a.correspondences()[1].view(np.uint32), b.correspondences()[1].view(np.uint32))
        assert rel_err(H2, H1) < 1e-4 and rel_err(b2, b1) < 1e-3 and abs(c2 - c1) < 1e-4 * c1
        assert rel_err(H2, H1) > 1e-9                                         # (the flag does reach the kernel)
        assert rel_err(b.mahalanobis()[:128], a.mahalanobis()[:128]) < 1e-3
        Ta, Tb, To = a.align(guess), b.align(guess), o.align(guess)
        te, re_ = scene.pose_error(Ta, Tb)
        print("fp32 per-point mode vs default:", te, re_, "counts", info_of(a), info_of(b))
        assert te <= 1e-5 and re_ <= 1e-6
        te, re_ = scene.pose_error(To, Tb)
        assert te <= T_TOL and re_ <= R_TOL


def test_algebraic_apd_is_an_opt_in_inside_the_tolerance(reg, golden, scene):
    """APDGICP_FLAG_ALGEBRAIC_APD (A:167-184 without atan2f / sincos): the search is untouched -- same correspondences
    and fp32 distances at a pose; M / H / b / cost equal the CHECKER OF THE MODE (oracle flags bit 3: the same ratios in libm arithmetic) to
    1e-10, and differ from the default's only by the fp32 rounding of the reference's three angles (< 1e-5 relative, > 1e-12: the flag does
    reach the kernel); final poses against the ORACLE'S DEFAULT (the reference's arithmetic) inside the north-star tolerance -- asserted a
    thousand times tighter; the flag is exclusive with the fp32 per-point mode."""
    src, tgt, guess = golden["lin_source"], golden["lin_target"], golden["lin_guess"]
    ALG = reg.FLAG_ALGEBRAIC_APD
    for xf in (0, 2):
        for kw in (LAUNCH, {}, dict(optimizer=1, max_iterations=20, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0)):
            a = reg.FastAPDGICP(reg.default_params(flags=xf, **kw))
            b = reg.FastAPDGICP(reg.default_params(flags=xf | ALG, **kw))
            o = R.RefAPDGICP(R.default_params(flags=xf, **kw))
            oa = R.RefAPDGICP(R.default_params(flags=xf | 8, **kw))
            for x in (a, b, o, oa):
                x.setInputSource(src)
                x.setInputTarget(tgt)
            T0 = guess.astype(np.float64)
            c1, H1, b1 = a.linearize(T0)
            c2, H2, b2 = b.linearize(T0)
            c3, H3, b3 = oa.linearize(T0)
            assert np.array_equal(a.correspondences()[0], b.correspondences()[0])
            assert np.array_equal(a.correspondences()[1].view(np.uint32), b.correspondences()[1].view(np.uint32))
            assert rel_err(H2, H3) < HB_TOL and rel_err(b2, b3) < HB_TOL and abs(c2 - c3) < HB_TOL * c3
            assert rel_err(b.mahalanobis(), oa.mahalanobis()) < HB_TOL
            e2, e3 = b.compute_error(T0), oa.compute_error(T0)
            assert abs(e2 - e3) < HB_TOL * e3
            assert 1e-12 < rel_err(H2, H1) < 1e-5 and rel_err(b2, b1) < 1e-4 and abs(c2 - c1) < 1e-5 * c1
            Ta, Tb, To, Toa = a.align(guess), b.align(guess), o.align(guess), oa.align(guess)
            assert info_of(b) == [int(oa.converged), oa.nr_iterations, oa.n_linearize, oa.n_compute_error]
            te, re_ = scene.pose_error(Toa, Tb)
            assert te <= 1e-9 and re_ <= 1e-9, (te, re_)
            te, re_ = scene.pose_error(To, Tb)
            print("algebraic sensor model vs the oracle's default arithmetic:", te, re_, "counts", info_of(a), info_of(b))
            assert te <= 1e-3 * T_TOL and re_ <= 5e-2 * R_TOL, (te, re_)   # (observed 3e-8 m / 1.3e-6 rad: a few ulps of the fp32 result matrix)
    with pytest.raises(reg.ApdgicpError):
        reg.FastAPDGICP(reg.default_params(flags=ALG | reg.FLAG_FP32_POINT_MATH))


def test_algebraic_apd_on_axis_points_follow_the_atan2_conventions(reg):
    """Points exactly on the sensor's axes, where the ratios are 0 / 0: the origin (the model vanishes), the z axis (azimuth = atan2(0, 0)
    = 0), the x axis (1 / cos(AoA) clamped to what the reference's fp32 pi/2 gives) -- the mode's kernel equals its checker and stays finite."""
    rng = np.random.default_rng(5)
    tgt = rng.uniform(-20, 20, size=(600, 3)).astype(np.float32)
    tgt[:8] = [[0, 0, 0], [0, 0, 7], [0, 0, -7], [9, 0, 0], [-9, 0, 0], [0, 4, 0], [1e-30, 0, 3], [5, 1e-38, 0]]
    src = tgt.copy()
    kw = dict(max_correspondence_distance=1.0, flags=reg.FLAG_ALGEBRAIC_APD)
    g = reg.FastAPDGICP(reg.default_params(**kw))
    o = R.RefAPDGICP(R.default_params(**dict(kw, flags=8)))
    for x in (g, o):
        x.setInputSource(src)
        x.setInputTarget(tgt)
    c1, H1, b1 = g.linearize(np.eye(4))
    c2, H2, b2 = o.linearize(np.eye(4))
    assert np.array_equal(g.correspondences()[0][:8], np.arange(8))
    assert np.isfinite(H1).all() and np.isfinite(g.mahalanobis()).all()
    assert rel_err(g.mahalanobis()[:8], o.mahalanobis()[:8]) < 1e-9
    assert rel_err(H1, H2) < 1e-9


def test_algebraic_apd_sweep_and_bench_pairs(reg, scene):
    """The mode over seeded small pairs (odometry and loop, LM with the launch parameters and GN-20) and four 8k bench pairs against the
    ORACLE'S DEFAULT arithmetic: every pose inside 1e-3 m / 1e-4 rad (observed ~1e-7 m); prints the share of LM pairs whose iteration
    count changes.  (tests/measure/algebraic_apd.py runs the 300-pair sweep and the 32 bench pairs: profiles/r06_algebraic_apd.json.)"""
    worst_t = worst_r = 0.0
    changed = total = 0
    GN = dict(optimizer=1, max_iterations=20, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0, azimuth_variance_deg=1.0)
    cases = [(kind, 1500 + 97 * i, 1400 + 131 * i, scene.pair_seed(7, i), kw) for i, (kind, kw) in enumerate([("odometry", LAUNCH), ("loop", LAUNCH), ("odometry", GN)] * 8)]
    cases += [("odometry", 8192, 8192, scene.pair_seed(2, p), GN) for p in range(2)] + [("loop", 8192, 8192, scene.pair_seed(4, p), LAUNCH) for p in range(2)]
    for kind, n, m, seed, kw in cases:
        src, tgt, _, guess = scene.make_pair(n, m, seed, kind)
        if kind == "loop":
            guess = np.eye(4, dtype=np.float32)
        g = reg.FastAPDGICP(reg.default_params(flags=reg.FLAG_ALGEBRAIC_APD, **kw))
        o = R.RefAPDGICP(R.default_params(**kw))
        for x in (g, o):
            x.setInputSource(src)
            x.setInputTarget(tgt)
        T, To = g.align(guess), o.align(guess)
        te, re_ = scene.pose_error(To, T)
        worst_t, worst_r = max(worst_t, te), max(worst_r, re_)
        assert te <= T_TOL and re_ <= R_TOL, (kind, n, te, re_)
        if kw is LAUNCH:
            total += 1
            changed += int(g.result.n_linearize != o.n_linearize)
    print(f"algebraic sensor model, {len(cases)} pairs vs the oracle's default: max {worst_t:.3e} m / {worst_r:.3e} rad; LM iteration count changed in {changed} of {total}")


def test_so3_exp_small_angle_branch_in_isolation(reg, golden, scene):
    """a15: so3_exp's Taylor branch (theta^2 < 1e-10, so3.hpp:63-68) -- a Gauss-Newton step of a few microradians: the source is the
    target moved by a rotation of 3e-6 rad and a micrometre, so the FIRST step's rotation vector is ~3e-6 (theta^2 ~ 1e-11: the
    Taylor branch), and the steps after it are smaller still.  Device state machine and host-driven loop against the oracle's
    trace, pose by pose; the branch that ran is read off the pose itself."""
    tgt = golden["lin_target"]
    Tt = scene.make_transform(np.array([1e-6, -2e-6, 0.5e-6]), 3e-6, -1e-6, 2e-6)
    Ti = np.linalg.inv(Tt)
    src = (tgt.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
    kw = dict(optimizer=1, max_iterations=3, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=1.0)
    o = R.RefAPDGICP(R.default_params(**kw))
    o.setInputSource(src)
    o.setInputTarget(tgt)
    o.align(None)
    want = o.trace()["poses"]
    assert len(want) == 3
    ang = [float(np.arccos(np.clip((np.trace(P[:3, :3]) - 1) / 2, -1, 1))) for P in want]
    assert 1e-7 < ang[0] < 1e-5, ang     # theta^2 < 1e-10: the Taylor branch produced this rotation
    for host_loop in (False, True):
        g = reg.FastAPDGICP(reg.default_params(**kw))
        g.setTrace(True)
        g.setInputSource(src)
        g.setInputTarget(tgt)
        g.align(None, host_loop=host_loop)
        got = g.trace()["poses"]
        assert got.shape == want.shape and np.abs(got - want).max() <= 1e-13, np.abs(got - want).max()
        assert np.abs(got[0][:3, :3] - np.eye(3)).max() < 1e-5 and np.abs(got[0][:3, :3] - np.eye(3)).max() > 1e-8
