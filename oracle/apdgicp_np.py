"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of RIV-SLAM's APD-GICP.

PARITY UNPINNED: the reference (fast_gicp::FastAPDGICP) needs PCL + Eigen + FLANN, none of which
exist in the build image, and the reference's own tests never instantiate FastAPDGICP
(fast_apdgicp/src/test/gicp_test.cpp:103-124).  Parity is therefore pinned by two independent
restatements that must agree with each other: this file (brute-force fp32 nearest neighbours,
numpy.linalg) and oracle/apdgicp_ref.cpp (kd-tree, hand-rolled linear algebra, OpenMP).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Every function cites the reference lines it follows; paths are relative to
/root/reference/fast_apdgicp/include/fast_gicp/.

Third-party arithmetic restated here (not vendored by the reference, versions unpinned there):
  * FLANN L2_Simple<float> distance as used by pcl::search::KdTree: result += (a[i]-b[i])^2 for
    i = 0,1,2 accumulated in fp32, no FMA (reference builds with -msse4.2 only,
    fast_apdgicp/CMakeLists.txt:11-16); exact k-NN.  Ties are broken towards the LOWER index here
    (FLANN's choice depends on tree layout and is not specified).
  * Eigen Isometry3f * Vector4f: row_i = (m_i0*x + m_i1*y) + (m_i2*z + m_i3*w) in fp32 as Eigen >= 3.3 sums it (the linear
    chain of Eigen 3.2 with Params.flags bit 1).
  * Eigen JacobiSVD of a symmetric PSD 3x3 == symmetric eigendecomposition, values descending.
  * Eigen Matrix4d::inverse() of blkdiag(C,1) == blkdiag(inv(C),1).
  * Eigen LDLT<6x6>::solve == any backward-stable SPD solve (agreement ~1e-12 relative).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

F32 = np.float32

# gicp/gicp_settings.hpp:6
REG_NONE, REG_MIN_EIG, REG_NORMALIZED_MIN_EIG, REG_PLANE, REG_FROBENIUS = 0, 1, 2, 3, 4
OPT_LM, OPT_GN = 0, 1


@dataclass
class Params:
    # gicp/impl/fast_apdgicp_impl.hpp:14-28, gicp/fast_apdgicp.hpp:107-109,
    # gicp/impl/lsq_registration_impl.hpp:11-24
    k_correspondences: int = 20
    max_iterations: int = 64
    lm_max_iterations: int = 10
    optimizer: int = OPT_LM
    regularization: int = REG_PLANE
    flags: int = 0   # bit 0: plain GICP (cov_dist omitted), gicp/impl/fast_gicp_impl.hpp; bit 1: T*p as a linear chain (Eigen 3.2); bit 3: the product's opt-in algebraic sensor model (not the reference's)
    max_correspondence_distance: float = float(np.finfo(np.float32).max)
    transformation_epsilon: float = 5e-4
    rotation_epsilon: float = 2e-3
    lm_init_lambda_factor: float = 1e-9
    distance_variance: float = 0.86
    azimuth_variance_deg: float = 0.5
    elevation_variance_deg: float = 1.0


# ----------------------------------------------------------------------------- fp32 geometry
def transform_points_f32(T: np.ndarray, pts: np.ndarray, linear_chain: bool = False) -> np.ndarray:
    """pt = trans.cast<float>() * p  (gicp/impl/fast_apdgicp_impl.hpp:137,149), fp32, no FMA.  The summation order is Eigen's:
    pairwise (r0 x + r1 y) + (r2 z + t) for Eigen >= 3.3 (redux_novec_unroller halves the four products of a row), the linear
    chain ((r0 x + r1 y) + r2 z) + t for Eigen 3.2 (Params.flags bit 1); see oracle/apdgicp_ref.cpp:xf_row."""
    Tf = np.asarray(T, dtype=np.float64).astype(F32)
    x, y, z = pts[:, 0].astype(F32), pts[:, 1].astype(F32), pts[:, 2].astype(F32)
    out = np.empty((pts.shape[0], 3), dtype=F32)
    for r in range(3):
        a = Tf[r, 0] * x
        a = a + Tf[r, 1] * y
        c = Tf[r, 2] * z
        out[:, r] = (a + c) + Tf[r, 3] if linear_chain else a + (c + Tf[r, 3])
    return out


def sqdist_f32(q: np.ndarray, t: np.ndarray) -> np.ndarray:
    """[len(q), len(t)] fp32 squared distances in FLANN L2_Simple accumulation order."""
    dx = q[:, None, 0] - t[None, :, 0]
    d = dx * dx
    dy = q[:, None, 1] - t[None, :, 1]
    d = d + dy * dy
    dz = q[:, None, 2] - t[None, :, 2]
    d = d + dz * dz
    return d


def nn1(q: np.ndarray, t: np.ndarray, chunk: int = 512):
    """Exact 1-NN (lowest index on ties).  Returns (idx int32, sqdist f32)."""
    idx = np.empty(q.shape[0], dtype=np.int32)
    sq = np.empty(q.shape[0], dtype=F32)
    for s in range(0, q.shape[0], chunk):
        d = sqdist_f32(q[s:s + chunk], t)
        j = np.argmin(d, axis=1)
        idx[s:s + chunk] = j
        sq[s:s + chunk] = d[np.arange(d.shape[0]), j]
    return idx, sq


def knn(cloud: np.ndarray, k: int, chunk: int = 512) -> np.ndarray:
    """Exact k-NN of every point within its own cloud, the point itself included
    (gicp/impl/fast_apdgicp_impl.hpp:316).  Ordered by (sqdist, index)."""
    n = cloud.shape[0]
    out = np.empty((n, k), dtype=np.int32)
    for s in range(0, n, chunk):
        d = sqdist_f32(cloud[s:s + chunk], cloud)
        out[s:s + chunk] = np.argsort(d, axis=1, kind="stable")[:, :k]
    return out


# ----------------------------------------------------------------------------- fp32 atan2 (the C library's)
def _f32(v):
    return np.asarray(v, dtype=F32)


def atanf_fdlibm(x: np.ndarray) -> np.ndarray:
    """glibc's generic flt-32 atanf (fdlibm s_atanf.c) on an fp32 array, every operation in fp32 in the published order:
    reduction to one of five intervals (|x| < 7/16, < 11/16, < 19/16, < 39/16, above), atan(c) as hi + lo for c = 0.5, 1, 1.5, inf,
    odd polynomial of degree 23 split into even and odd powers.  A second, independent writing of include/apd_atan2f.h
    (tests/test_atan2f.py compares the two, and both with the C library, bit for bit)."""
    x = _f32(x)
    ix = x.view(np.int32) & 0x7FFFFFFF
    neg = x.view(np.int32) < 0
    ax = np.abs(x)
    hi = _f32([4.6364760399e-01, 7.8539812565e-01, 9.8279368877e-01, 1.5707962513e+00])
    lo = _f32([5.0121582440e-09, 3.7748947079e-08, 3.4473217170e-08, 7.5497894159e-08])
    aT = _f32([3.3333334327e-01, -2.0000000298e-01, 1.4285714924e-01, -1.1111110449e-01, 9.0908870101e-02, -7.6918758452e-02,
               6.6610731184e-02, -5.8335702866e-02, 4.9768779427e-02, -3.6531571299e-02, 1.6285819933e-02])
    one, two, onep5 = F32(1.0), F32(2.0), F32(1.5)
    idn = np.where(ix < 0x3EE00000, -1, np.where(ix < 0x3F300000, 0, np.where(ix < 0x3F980000, 1, np.where(ix < 0x401C0000, 2, 3))))
    with np.errstate(all="ignore"):
        r = np.where(idn == -1, x,
                     np.where(idn == 0, (two * ax - one) / (two + ax),
                              np.where(idn == 1, (ax - one) / (ax + one),
                                       np.where(idn == 2, (ax - onep5) / (one + onep5 * ax), -one / ax)))).astype(F32)
        z = r * r
        w = z * z
        s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))))
        s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))))
        small = r - r * (s1 + s2)
        k = np.clip(idn, 0, 3)
        big = hi[k] - ((r * (s1 + s2) - lo[k]) - r)
        out = np.where(idn < 0, small, np.where(neg, -big, big)).astype(F32)
        out = np.where(ix < 0x31000000, x, out)                                   # |x| < 2^-29
        huge = np.where(neg, -hi[3] - lo[3], hi[3] + lo[3]).astype(F32)
        out = np.where(ix >= 0x4C000000, np.where(ix > 0x7F800000, x + x, huge), out)  # |x| >= 2^25, NaN
    return out.astype(F32)


def atan2f_fdlibm(y: np.ndarray, x: np.ndarray) -> np.ndarray:
    """glibc's generic flt-32 atan2f (fdlibm e_atan2f.c): what `atan2(float, float)` is at fast_apdgicp_impl.hpp:168,172-173 on
    the platforms the reference names (glibc 2.27 / 2.31)."""
    y, x = np.broadcast_arrays(_f32(y), _f32(x))
    y, x = np.ascontiguousarray(y), np.ascontiguousarray(x)
    hx, hy = x.view(np.int32), y.view(np.int32)
    ix, iy = hx & 0x7FFFFFFF, hy & 0x7FFFFFFF
    pi_o_4, pi_o_2, pi, pi_lo = F32(7.8539818525e-01), F32(1.5707963705e+00), F32(3.1415927410e+00), F32(-8.7422776573e-08)
    m = ((hy >> 31) & 1) | ((hx >> 30) & 2)
    k = (iy - ix) >> 23
    with np.errstate(all="ignore"):
        z = atanf_fdlibm(np.abs(y / x))
        z = np.where(k > 60, pi_o_2 + F32(0.5) * pi_lo, np.where((hx < 0) & (k < -60), F32(0.0), z)).astype(F32)
        out = np.where(m == 0, z, np.where(m == 1, -z, np.where(m == 2, pi - (z - pi_lo), (z - pi_lo) - pi))).astype(F32)
        # special cases, in the order of the original (the later line wins here, so they are applied last to first)
        y_inf = np.where(hy < 0, -pi_o_2, pi_o_2)
        out = np.where(iy == 0x7F800000, y_inf, out)
        x_inf_y_inf = np.where(m == 0, pi_o_4, np.where(m == 1, -pi_o_4, np.where(m == 2, F32(3.0) * pi_o_4, F32(-3.0) * pi_o_4)))
        x_inf = np.where(m == 0, F32(0.0), np.where(m == 1, F32(-0.0), np.where(m == 2, pi, -pi)))
        out = np.where(ix == 0x7F800000, np.where(iy == 0x7F800000, x_inf_y_inf, x_inf), out)
        out = np.where(ix == 0, np.where(hy < 0, -pi_o_2, pi_o_2), out)
        out = np.where(iy == 0, np.where(m < 2, y, np.where(m == 2, pi, -pi)), out)
        out = np.where(hx == 0x3F800000, atanf_fdlibm(y), out)
        out = np.where((ix > 0x7F800000) | (iy > 0x7F800000), x + y, out)
    return out.astype(F32)


# ----------------------------------------------------------------------------- covariances
def calculate_covariances(cloud: np.ndarray, k: int = 20, regularization: int = REG_PLANE) -> np.ndarray:
    """gicp/impl/fast_apdgicp_impl.hpp:303-363.  Returns [n,3,3] fp64 (top-left block of the
    reference's Matrix4d; its 4th row/column is identically zero)."""
    cloud = np.ascontiguousarray(cloud, dtype=F32)
    n = cloud.shape[0]
    if n < k:
        raise ValueError("need at least k points (reference reads uninitialised neighbours, :318-321)")
    nbr = knn(cloud, k)
    covs = np.empty((n, 3, 3), dtype=np.float64)
    for i in range(n):
        P = cloud[nbr[i]].astype(np.float64)           # :318-321
        P = P - P.mean(axis=0)                          # :323
        cov = (P.T @ P) / k                             # :324  (population, /k)
        if regularization == REG_NONE:                  # :326-328
            covs[i] = cov
        elif regularization == REG_FROBENIUS:           # :329-335
            C = cov + 1e-3 * np.eye(3)
            Ci = np.linalg.inv(C)
            covs[i] = np.linalg.inv(Ci / np.linalg.norm(Ci))
        else:                                           # :337-357
            w, U = np.linalg.eigh(cov)
            w, U = w[::-1], U[:, ::-1]                  # singular values descending
            w = np.maximum(w, 0.0)
            if regularization == REG_PLANE:
                vals = np.array([1.0, 1.0, 1e-3])
            elif regularization == REG_MIN_EIG:
                vals = np.maximum(w, 1e-3)
            elif regularization == REG_NORMALIZED_MIN_EIG:
                vals = np.maximum(w / w.max(), 1e-3)
            else:
                raise ValueError("unknown regularization (reference aborts, :341-343)")
            covs[i] = (U * vals) @ U.T
    return covs


# ----------------------------------------------------------------------------- SO(3)
def skewd(x):
    """so3/so3.hpp:21-31"""
    return np.array([[0, -x[2], x[1]], [x[2], 0, -x[0]], [-x[1], x[0], 0]], dtype=np.float64)


def so3_exp(omega: np.ndarray) -> np.ndarray:
    """so3/so3.hpp:59-78 -> unit quaternion -> Eigen::Quaterniond::toRotationMatrix()."""
    theta_sq = float(omega @ omega)
    if theta_sq < 1e-10:
        theta_quad = theta_sq * theta_sq
        imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad
        real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad
    else:
        theta = math.sqrt(theta_sq)
        half = 0.5 * theta
        imag = math.sin(half) / theta
        real = math.cos(half)
    w, x, y, z = real, imag * omega[0], imag * omega[1], imag * omega[2]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1 - (txx + tyy)]], dtype=np.float64)


# ----------------------------------------------------------------------------- the registration
@dataclass
class Trace:
    lambdas: list = field(default_factory=list)
    rhos: list = field(default_factory=list)
    y0s: list = field(default_factory=list)
    yis: list = field(default_factory=list)
    poses: list = field(default_factory=list)   # x0 after every outer iteration
    n_linearize: int = 0
    n_compute_error: int = 0


class FastAPDGICP:
    """Mirror of fast_gicp::FastAPDGICP + LsqRegistration (method names follow the reference)."""

    def __init__(self, params: Params | None = None):
        self.p = params or Params()
        self.source = None
        self.target = None
        self.source_covs = None
        self.target_covs = None
        self.correspondences = None
        self.sq_distances = None
        self.mahalanobis = None
        self.final_transformation = np.eye(4, dtype=F32)
        self.final_hessian = np.eye(6)
        self.converged = False
        self.nr_iterations = 0
        self.lm_lambda = -1.0
        self.trace = Trace()

    # gicp/impl/fast_apdgicp_impl.hpp:90-108 (pointer-equality caching is the adapter's business)
    def setInputSource(self, cloud):
        self.source = np.ascontiguousarray(cloud, dtype=F32)
        self.source_covs = None

    def setInputTarget(self, cloud):
        self.target = np.ascontiguousarray(cloud, dtype=F32)
        self.target_covs = None

    # :133-194
    def update_correspondences(self, T: np.ndarray):
        p = self.p
        pt = transform_points_f32(T, self.source, bool(p.flags & 2))
        idx, sq = nn1(pt, self.target)
        thr2 = float(p.max_correspondence_distance) * float(p.max_correspondence_distance)
        corr = np.where(sq.astype(np.float64) < thr2, idx, -1).astype(np.int32)  # :156
        n = self.source.shape[0]
        M = np.zeros((n, 3, 3))
        R = np.asarray(T, dtype=np.float64)[:3, :3]
        sin_az = math.sin(p.azimuth_variance_deg / 180 * math.pi)
        sin_el = math.sin(p.elevation_variance_deg / 180 * math.pi)
        # the three angles (:168,172-173): float overloads of atan2 / sqrt on fp32 members -- the C library's atan2f, restated
        px, py, pz = pt[:, 0], pt[:, 1], pt[:, 2]
        aoa_f = atan2f_fdlibm(px, np.sqrt(py * py + pz * pz))
        elev_f = atan2f_fdlibm(np.sqrt(px * px + py * py), pz)
        azim_f = atan2f_fdlibm(py, px)
        for i in range(n):
            j = corr[i]
            if j < 0:
                continue
            dist = float(np.linalg.norm(pt[i].astype(np.float64)))            # :167
            aoa = float(aoa_f[i])                                              # :168 (float overloads)
            s_x = dist * p.distance_variance / 400                             # :169
            s_y = dist * sin_az / math.cos(aoa)                                # :170
            s_z = dist * sin_el / math.cos(aoa)                                # :171
            elevation = float(elev_f[i])                                       # :172
            azimuth = float(azim_f[i])                                         # :173
            ce, se = math.cos(elevation), math.sin(elevation)
            ca, sa = math.cos(azimuth), math.sin(azimuth)
            if p.flags & 8:
                # NOT the reference: the checker of the product's opt-in APDGICP_FLAG_ALGEBRAIC_APD -- the same ratios from the coordinates
                x, y, z = (float(v) for v in pt[i])
                rho, yz = math.sqrt(x * x + y * y), math.sqrt(y * y + z * z)
                inv_cos = min(dist / yz if yz > 0 else math.inf, 1.0 / 4.371138828673793e-08)
                s_y, s_z = dist * inv_cos * sin_az, dist * inv_cos * sin_el
                se, ce = (rho / dist, z / dist) if dist > 0 else (0.0, 0.0)
                ca, sa = (x / rho, y / rho) if rho > 0 else (1.0, 0.0)
            Ry = np.array([[ce, 0, se], [0, 1, 0], [-se, 0, ce]])
            Rz = np.array([[ca, -sa, 0], [sa, ca, 0], [0, 0, 1.0]])
            A = (Rz @ Ry) @ np.diag([s_x, s_y, s_z])                           # :174-181
            cov_r = A @ A.T                                                    # :182
            if p.flags & 1:
                cov_r = np.zeros((3, 3))                                       # upstream FastGICP
            RCR = (self.target_covs[j] + cov_r) + R @ (self.source_covs[i] + cov_r) @ R.T  # :188
            M[i] = np.linalg.inv(RCR)                                          # :191-192
        self.correspondences, self.sq_distances, self.mahalanobis = corr, sq, M

    # :198-272
    def linearize(self, T: np.ndarray, want_Hb: bool = True):
        self.trace.n_linearize += 1
        self.update_correspondences(T)
        T = np.asarray(T, dtype=np.float64)
        sel = np.nonzero(self.correspondences >= 0)[0]
        a = self.source[sel].astype(np.float64)
        b = self.target[self.correspondences[sel]].astype(np.float64)
        Ta = a @ T[:3, :3].T + T[:3, 3]
        e = b - Ta
        M = self.mahalanobis[sel]
        Me = np.einsum("nij,nj->ni", M, e)
        cost = float(np.einsum("ni,ni->", e, Me))
        if not want_Hb:
            return cost, None, None
        J = np.zeros((len(sel), 3, 6))
        J[:, 0, 1], J[:, 0, 2] = -Ta[:, 2], Ta[:, 1]
        J[:, 1, 0], J[:, 1, 2] = Ta[:, 2], -Ta[:, 0]
        J[:, 2, 0], J[:, 2, 1] = -Ta[:, 1], Ta[:, 0]
        J[:, 0, 3] = J[:, 1, 4] = J[:, 2, 5] = -1.0
        MJ = np.einsum("nij,njk->nik", M, J)
        H = np.einsum("nji,njk->ik", J, MJ)
        bb = np.einsum("nji,nj->i", J, Me)
        return cost, H, bb

    # :275-298
    def compute_error(self, T: np.ndarray) -> float:
        self.trace.n_compute_error += 1
        T = np.asarray(T, dtype=np.float64)
        sel = np.nonzero(self.correspondences >= 0)[0]
        a = self.source[sel].astype(np.float64)
        b = self.target[self.correspondences[sel]].astype(np.float64)
        e = b - (a @ T[:3, :3].T + T[:3, 3])
        return float(np.einsum("ni,nij,nj->", e, self.mahalanobis[sel], e))

    # gicp/impl/lsq_registration_impl.hpp:83-92
    def is_converged(self, delta: np.ndarray) -> bool:
        with np.errstate(divide="ignore", invalid="ignore"):
            r = np.abs(delta[:3, :3] - np.eye(3)) * (1.0 / self.p.rotation_epsilon)
            t = np.abs(delta[:3, 3]) * (1.0 / self.p.transformation_epsilon)
        return max(r.max(), t.max()) < 1

    @staticmethod
    def _delta(d: np.ndarray) -> np.ndarray:
        delta = np.eye(4)
        delta[:3, :3] = so3_exp(d[:3])
        delta[:3, 3] = d[3:]
        return delta

    # lsq_registration_impl.hpp:107-123
    def step_gn(self, x0):
        y0, H, b = self.linearize(x0)
        d = np.linalg.solve(H, -b)
        delta = self._delta(d)
        self.final_hessian = H
        self.trace.y0s.append(y0)
        return True, delta @ x0, delta

    # lsq_registration_impl.hpp:127-173
    def step_lm(self, x0):
        y0, H, b = self.linearize(x0)
        if self.lm_lambda < 0.0:
            self.lm_lambda = self.p.lm_init_lambda_factor * np.abs(np.diag(H)).max()
        nu = 2.0
        delta = np.eye(4)
        for _ in range(self.p.lm_max_iterations):
            d = np.linalg.solve(H + self.lm_lambda * np.eye(6), -b)
            delta = self._delta(d)
            xi = delta @ x0
            yi = self.compute_error(xi)
            rho = (y0 - yi) / float(d @ (self.lm_lambda * d - b))
            self.trace.lambdas.append(self.lm_lambda)
            self.trace.rhos.append(rho)
            self.trace.y0s.append(y0)
            self.trace.yis.append(yi)
            if rho < 0:
                if self.is_converged(delta):
                    return True, x0, delta
                self.lm_lambda = nu * self.lm_lambda
                nu = 2 * nu
                continue
            self.lm_lambda = self.lm_lambda * max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3)
            self.final_hessian = H
            return True, xi, delta
        return False, x0, delta

    # fast_apdgicp_impl.hpp:121-130 + lsq_registration_impl.hpp:55-80
    def align(self, guess=None):
        p = self.p
        if self.source_covs is None:
            self.source_covs = calculate_covariances(self.source, p.k_correspondences, p.regularization)
        if self.target_covs is None:
            self.target_covs = calculate_covariances(self.target, p.k_correspondences, p.regularization)
        g = np.eye(4, dtype=F32) if guess is None else np.asarray(guess, dtype=F32)
        x0 = g.astype(np.float64)
        self.lm_lambda = -1.0
        self.converged = False
        self.trace = Trace()
        self.nr_iterations = 0
        for i in range(p.max_iterations):
            if self.converged:
                break
            self.nr_iterations = i
            ok, x0, delta = (self.step_lm if p.optimizer == OPT_LM else self.step_gn)(x0)
            if not ok:
                break  # "lm not converged!!"
            self.converged = self.is_converged(delta)
            self.trace.poses.append(x0.copy())
        self.final_transformation = x0.astype(F32)
        return self.final_transformation

    def hasConverged(self):
        return self.converged

    def getFinalTransformation(self):
        return self.final_transformation
