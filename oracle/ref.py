"""ORACLE loader (test infrastructure): ctypes binding of oracle/_build/libapdgicp_ref.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libapdgicp_ref.so")


class RefParams(C.Structure):
    _fields_ = [
        ("k_correspondences", C.c_int32),
        ("max_iterations", C.c_int32),
        ("lm_max_iterations", C.c_int32),
        ("optimizer", C.c_int32),
        ("regularization", C.c_int32),
        ("flags", C.c_int32),
        ("max_correspondence_distance", C.c_double),
        ("transformation_epsilon", C.c_double),
        ("rotation_epsilon", C.c_double),
        ("lm_init_lambda_factor", C.c_double),
        ("distance_variance", C.c_double),
        ("azimuth_variance_deg", C.c_double),
        ("elevation_variance_deg", C.c_double),
    ]


def default_params(**kw) -> RefParams:
    p = RefParams(20, 64, 10, 0, 3, 0, float(np.finfo(np.float32).max), 5e-4, 2e-3, 1e-9, 0.86, 0.5, 1.0)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def build(force: bool = False) -> str:
    import fcntl
    srcs = [os.path.join(_HERE, "apdgicp_ref.cpp"), os.path.join(_HERE, "..", "include", "apd_atan2f.h"), os.path.join(_HERE, "Makefile")]

    def stale():
        return force or not os.path.exists(_LIB_PATH) or any(os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale():
        os.makedirs(os.path.dirname(_LIB_PATH), exist_ok=True)
        with open(_LIB_PATH + ".lock", "w") as lock:  # one builder at a time (bench.py: every rank of a node gets here)
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if stale():
                    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.ref_create.restype = C.c_void_p
        L.ref_create.argtypes = [C.POINTER(RefParams)]
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_set_params.argtypes = [C.c_void_p, C.POINTER(RefParams)]
        L.ref_set_num_threads.argtypes = [C.c_void_p, C.c_int]
        for f in (L.ref_set_source, L.ref_set_target):
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
            f.restype = None
        L.ref_compute_covariances.argtypes = [C.c_void_p, C.c_int]
        L.ref_get_covariances.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ref_linearize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.ref_compute_error.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.ref_get_correspondences.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_get_mahalanobis.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_align.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_get_final_hessian.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_get_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.ref_atan2f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]
        L.ref_atan2f.restype = None
        L.ref_knn_bruteforce.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_knn_kdtree.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_knn_kdtree_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_submap_assemble.restype = C.c_longlong
        L.ref_submap_assemble.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class RefAPDGICP:
    """The C++ restatement behind the reference's method names (matrices are numpy row-major
    [4,4] / [6,6] views; the C side is column-major like Eigen)."""

    def __init__(self, params: RefParams | None = None, num_threads: int = 0):
        self.L = lib()
        self.params = params or default_params()
        self.h = C.c_void_p(self.L.ref_create(C.byref(self.params)))
        self.num_threads = self.L.ref_set_num_threads(self.h, num_threads)
        self.n_src = self.n_tgt = 0
        self.converged = False
        self.nr_iterations = 0
        self.n_linearize = self.n_compute_error = 0

    def __del__(self):
        try:
            if self.h:
                self.L.ref_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_params(self, params: RefParams):
        self.params = params
        self.L.ref_set_params(self.h, C.byref(params))

    def setInputSource(self, cloud):
        c = np.ascontiguousarray(cloud, dtype=np.float32)
        self._src = c
        self.n_src = c.shape[0]
        self.L.ref_set_source(self.h, _ptr(c), c.shape[0], c.shape[1])

    def setInputTarget(self, cloud):
        c = np.ascontiguousarray(cloud, dtype=np.float32)
        self._tgt = c
        self.n_tgt = c.shape[0]
        self.L.ref_set_target(self.h, _ptr(c), c.shape[0], c.shape[1])

    def covariances(self, which: str) -> np.ndarray:
        w = 0 if which == "source" else 1
        n = self.n_src if w == 0 else self.n_tgt
        rc = self.L.ref_compute_covariances(self.h, w)
        if rc:
            raise RuntimeError(f"calculate_covariances failed rc={rc}")
        out = np.empty((n, 3, 3))
        self.L.ref_get_covariances(self.h, w, _ptr(out))
        return out

    def linearize(self, T, want_Hb: bool = True):
        Tc = np.asfortranarray(np.asarray(T, dtype=np.float64))
        H = np.zeros((6, 6), order="F")
        b = np.zeros(6)
        cost = C.c_double()
        rc = self.L.ref_linearize(self.h, _ptr(Tc), _ptr(H) if want_Hb else None, _ptr(b) if want_Hb else None, C.byref(cost))
        if rc:
            raise RuntimeError("ref_linearize failed")
        return cost.value, (np.ascontiguousarray(H) if want_Hb else None), (b if want_Hb else None)

    def compute_error(self, T) -> float:
        Tc = np.asfortranarray(np.asarray(T, dtype=np.float64))
        cost = C.c_double()
        if self.L.ref_compute_error(self.h, _ptr(Tc), C.byref(cost)):
            raise RuntimeError("ref_compute_error failed (no correspondences yet)")
        return cost.value

    def correspondences(self):
        corr = np.empty(self.n_src, dtype=np.int32)
        sqd = np.empty(self.n_src, dtype=np.float32)
        self.L.ref_get_correspondences(self.h, _ptr(corr), _ptr(sqd))
        return corr, sqd

    def mahalanobis(self) -> np.ndarray:
        out = np.empty((self.n_src, 3, 3))
        self.L.ref_get_mahalanobis(self.h, _ptr(out))
        return out

    def align(self, guess=None) -> np.ndarray:
        g = np.eye(4, dtype=np.float32) if guess is None else np.asarray(guess, dtype=np.float32)
        gc = np.asfortranarray(g)
        out = np.zeros((4, 4), dtype=np.float32, order="F")
        info = np.zeros(4, dtype=np.int32)
        if self.L.ref_align(self.h, _ptr(gc), _ptr(out), _ptr(info)):
            raise RuntimeError("ref_align failed")
        self.converged = bool(info[0])
        self.nr_iterations = int(info[1])
        self.n_linearize, self.n_compute_error = int(info[2]), int(info[3])
        self.final_transformation = np.ascontiguousarray(out)
        return self.final_transformation

    def trace(self):
        """dict(lambda, rho, y0, yi, poses) of the last align: per LM trial the lambda it was solved with, its rho and the two costs
        rho compares (L:137-146); per completed outer iteration the pose behind it ([n, 4, 4] row-major numpy)."""
        cap = max(1, self.params.max_iterations) * max(1, self.params.lm_max_iterations)
        lam, rho, y0, yi = np.zeros(cap), np.zeros(cap), np.zeros(cap), np.zeros(cap)
        poses = np.zeros((max(1, self.params.max_iterations), 16))
        npo = C.c_int()
        nt = self.L.ref_get_trace(self.h, _ptr(lam), _ptr(rho), _ptr(y0), _ptr(yi), _ptr(poses), C.byref(npo))
        return {"lambda": lam[:nt].copy(), "rho": rho[:nt].copy(), "y0": y0[:nt].copy(), "yi": yi[:nt].copy(),
                "poses": poses[:npo.value].reshape(-1, 4, 4).transpose(0, 2, 1).copy()}

    def final_hessian(self) -> np.ndarray:
        H = np.zeros((6, 6), order="F")
        self.L.ref_get_final_hessian(self.h, _ptr(H))
        return np.ascontiguousarray(H)

    def hasConverged(self):
        return self.converged

    def getFinalTransformation(self):
        return self.final_transformation

    def knn_kdtree(self, which: str, q, k: int):
        idx = np.empty(k, dtype=np.int32)
        d = np.empty(k, dtype=np.float32)
        qq = np.ascontiguousarray(q, dtype=np.float32)
        n = self.L.ref_knn_kdtree(self.h, 0 if which == "source" else 1, _ptr(qq), k, _ptr(idx), _ptr(d))
        return idx[:n], d[:n]


    def knn_kdtree_batch(self, which: str, q, k: int):
        """k nearest neighbours of every row of q [nq, 3] through the restatement's kd-tree: (idx [nq, k], fp32 d2 [nq, k])."""
        qq = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, 3)
        idx = np.empty((len(qq), k), dtype=np.int32)
        d = np.empty((len(qq), k), dtype=np.float32)
        self.L.ref_knn_kdtree_batch(self.h, 0 if which == "source" else 1, _ptr(qq), len(qq), k, _ptr(idx), _ptr(d))
        return idx, d


REF_NANOFLANN = "/root/reference/radar_graph_slam/include/scan_context/nanoflann.hpp"


def nanoflann_lib():
    """oracle/_ref/libnanoflann_nn.so -- the reference tree's own header-only nanoflann (ScanContext module) behind a 40-line harness
    (oracle/nanoflann_harness.cpp), compiled from the header where it lies.  None when /root/reference is absent (the GPU box)."""
    path = os.path.join(_HERE, "_ref", "libnanoflann_nn.so")
    if os.path.exists(REF_NANOFLANN):
        subprocess.check_call(["make", "-C", _HERE, "-s", "_ref/libnanoflann_nn.so"])
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.nf_knn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.nf_knn.restype = C.c_int
    return L


def nanoflann_knn(cloud, q, k: int, leaf_max_size: int = 10):
    L = nanoflann_lib()
    assert L is not None
    cc = np.ascontiguousarray(cloud, dtype=np.float32).reshape(-1, 3)
    qq = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, 3)
    idx = np.empty((len(qq), k), dtype=np.int32)
    d = np.empty((len(qq), k), dtype=np.float32)
    L.nf_knn(_ptr(cc), len(cc), _ptr(qq), len(qq), k, leaf_max_size, _ptr(idx), _ptr(d))
    return idx, d


def atan2f(y, x) -> np.ndarray:
    """include/apd_atan2f.h (the C library's fdlibm atan2f restated) on fp32 arrays."""
    y = np.ascontiguousarray(y, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(y)
    lib().ref_atan2f(_ptr(y), _ptr(x), _ptr(out), y.size)
    return out


def submap_assemble(clouds, rel_poses=None, leaf=None):
    """scan_matching_odometry_nodelet.cpp:606-618 + downsample() (:412-422) on the CPU.
    clouds: list of [n, 4] float32 {x, y, z, intensity}; rel_poses: list of 4x4 (row-major numpy) or None;
    leaf: float or 3 floats or None.  Returns (points [m, 4] float32, voxel index [m], population [m])."""
    L = lib()
    cs = [np.ascontiguousarray(c, dtype=np.float32).reshape(-1, 4) for c in clouds]
    ptrs = (C.c_void_p * len(cs))(*[c.ctypes.data for c in cs])
    ns = np.array([c.shape[0] for c in cs], dtype=np.int64)
    poses = None
    if rel_poses is not None:
        poses = np.ascontiguousarray(np.stack([np.asarray(T, dtype=np.float64).T.reshape(-1) for T in rel_poses]))
    lf = None
    if leaf is not None:
        lf = np.ascontiguousarray(np.broadcast_to(np.asarray(leaf, dtype=np.float32), (3,)))
    total = int(ns.sum())
    out = np.empty((max(total, 1), 4), dtype=np.float32)
    idx = np.zeros(max(total, 1), dtype=np.int32)
    cnt = np.zeros(max(total, 1), dtype=np.int32)
    m = L.ref_submap_assemble(len(cs), ptrs, _ptr(ns), _ptr(poses) if poses is not None else None,
                              _ptr(lf) if lf is not None else None, _ptr(out), _ptr(idx), _ptr(cnt))
    if m < 0:   # PCL: "Leaf size is too small for the input dataset. Integer indices would overflow." -- a warning; output = input
        n = -m - 1
        return out[:n].copy(), np.full(n, -1, dtype=np.int32), np.ones(n, dtype=np.int32)
    return out[:m].copy(), idx[:m].copy(), cnt[:m].copy()
