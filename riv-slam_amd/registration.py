"""ctypes binding of libapdgicp_hip.so (include/apdgicp_hip.h) with the reference's method names.

`FastAPDGICP` mirrors fast_gicp::FastAPDGICP as seen through pcl::Registration
(fast_apdgicp/include/fast_gicp/gicp/fast_apdgicp.hpp:19-110 and the calls made by
radar_graph_slam/src/radar_graph_slam/registrations.cpp:38-50,
radar_graph_slam/apps/scan_matching_odometry_nodelet.cpp:437-482): setInputSource / setInputTarget /
align / hasConverged / getFinalTransformation / getFitnessScore plus the eight setters of the ROS
factory.  `BatchAPDGICP` is the batched entry (loop-closure candidates, loop_detector.cpp:222-236).

There is NO CPU fallback: if the HIP library or a GPU is missing every call raises.
numpy matrices are row-major [4,4]/[6,6]; the C ABI is column-major (Eigen) -- converted here.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libapdgicp_hip.so")

REG_NONE, REG_MIN_EIG, REG_NORMALIZED_MIN_EIG, REG_PLANE, REG_FROBENIUS = 0, 1, 2, 3, 4
OPT_LM, OPT_GN = 0, 1
FLAG_PLAIN_GICP = 1  # upstream fast_gicp::FastGICP cost (no APD covariance)
FLAG_XF_LINEAR_CHAIN = 2  # T*p summed ((r0 x + r1 y) + r2 z) + t like Eigen 3.2 instead of pairwise like Eigen >= 3.3 (include/apdgicp_hip.h)
FLAG_FP32_POINT_MATH = 4  # opt-in: the per-point algebra behind the search in fp32 (include/apdgicp_hip.h); not the reference's precision
FLAG_ALGEBRAIC_APD = 8  # opt-in: the sensor model's sines / cosines as coordinate ratios, no atan2f / sincos (include/apdgicp_hip.h); not the reference's arithmetic
SOURCE, TARGET = 0, 1


class ApdgicpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"apdgicp error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    """apdgicp_params (include/apdgicp_hip.h)."""
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


class Result(C.Structure):
    """apdgicp_result (include/apdgicp_hip.h)."""
    _fields_ = [
        ("T", C.c_float * 16),
        ("final_cost", C.c_double),
        ("converged", C.c_int32),
        ("iterations", C.c_int32),
        ("n_linearize", C.c_int32),
        ("n_compute_error", C.c_int32),
        ("lm_failed", C.c_int32),
        ("n_matched", C.c_int32),
    ]

    def matrix(self) -> np.ndarray:
        return np.array(self.T, dtype=np.float32).reshape(4, 4).T.copy()


class Pair(C.Structure):
    """apdgicp_pair"""
    _fields_ = [("source_cloud", C.c_int32), ("target_cloud", C.c_int32), ("guess", C.c_float * 16)]


RESULT_DTYPE = np.dtype([("T", np.float32, (16,)), ("final_cost", np.float64), ("converged", np.int32), ("iterations", np.int32),
                         ("n_linearize", np.int32), ("n_compute_error", np.int32), ("lm_failed", np.int32), ("n_matched", np.int32)])
assert RESULT_DTYPE.itemsize == C.sizeof(Result) == 96

# every symbol include/apdgicp_hip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "apdgicp_abi_version", "apdgicp_source_stamp", "apdgicp_build_flags", "apdgicp_last_error", "apdgicp_device_count", "apdgicp_default_params",
    "apdgicp_set_trace", "apdgicp_get_trace", "apdgicp_get_trace_step_norms", "apdgicp_debug_atan2f", "apdgicp_nearest_neighbours", "apdgicp_nearest_neighbours_of", "apdgicp_get_points",
    "apdgicp_create", "apdgicp_destroy", "apdgicp_set_params", "apdgicp_get_params",
    "apdgicp_set_source", "apdgicp_set_target", "apdgicp_clear_source", "apdgicp_clear_target",
    "apdgicp_swap_source_and_target", "apdgicp_compute_covariances", "apdgicp_get_covariances",
    "apdgicp_set_covariances", "apdgicp_linearize", "apdgicp_compute_error", "apdgicp_get_correspondences",
    "apdgicp_get_mahalanobis", "apdgicp_align", "apdgicp_align_host_loop", "apdgicp_get_final_hessian",
    "apdgicp_transform_source", "apdgicp_fitness_score", "apdgicp_inlier_fraction", "apdgicp_synchronize", "apdgicp_wait_producer",
    "apdgicp_batch_wait_producer", "apdgicp_get_stream", "apdgicp_batch_get_stream",
    "apdgicp_batch_create", "apdgicp_batch_destroy", "apdgicp_batch_set_params", "apdgicp_batch_clear",
    "apdgicp_batch_add_cloud", "apdgicp_batch_set_cloud", "apdgicp_batch_set_clouds", "apdgicp_batch_compute_covariances", "apdgicp_batch_align",
    "apdgicp_batch_align_async", "apdgicp_batch_pump", "apdgicp_batch_is_pooled", "apdgicp_batch_fitness", "apdgicp_batch_synchronize", "apdgicp_batch_copy_results", "apdgicp_batch_set_profiling",
    "apdgicp_batch_align_enqueue", "apdgicp_batch_align_collect", "apdgicp_batch_set_pair_groups", "apdgicp_batch_last_nn_time", "apdgicp_batch_last_nn_profile", "apdgicp_batch_last_ticks", "apdgicp_batch_last_nn_kernel", "apdgicp_batch_debug_stats", "apdgicp_batch_debug_block_timeline", "apdgicp_batch_pool_counters",
    "apdgicp_submap_create", "apdgicp_submap_destroy", "apdgicp_submap_assemble", "apdgicp_submap_points", "apdgicp_submap_copy",
]

_lib = None


def load_library(path: str | None = None):
    """Loads libapdgicp_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # torch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1.  Whichever copy is
    # loaded first serves the whole process, and two different ROCm runtimes in one process do not
    # work ("No HIP GPUs are available"), so when torch is installed let it load its runtime first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(p)
    vp, i32, i64, u64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double
    # The library says which sources it was compiled from (apdgicp_source_stamp); one that was built from other sources than the
    # ones beside it is refused, whatever its age (a prebuilt library pushed with the tree would otherwise be measured and tested
    # under the name of sources it does not contain).  APDGICP_ALLOW_STALE_LIB=1 is for bisecting with hand-built libraries.
    L.apdgicp_source_stamp.restype = C.c_char_p
    if path is None and os.environ.get("APDGICP_ALLOW_STALE_LIB", "0") != "1":
        from . import build as _build
        try:
            want = _build.source_stamp()
        except OSError:
            want = None   # a deployment that ships the library and this package without csrc/ and include/: nothing to compare with
        have = L.apdgicp_source_stamp().decode()
        if want is not None and have != want:
            raise RuntimeError(f"{p} was compiled from other sources (stamp {have}) than the ones on disk ({want}): rebuild it "
                               "(`python -c 'import __graft_entry__ as g; g.build()'`)")
    # ... and how it was compiled: a library that lists an experiment define (APD_ABL_*: ablations that are wrong by design; APD_OCML_ATAN2F,
    # APD_SINCOS_NO_TABLE: A/B builds) behind "variant:" is not the product, whatever its stamp says (tools/ab_bench.sh sets the override)
    try:
        L.apdgicp_build_flags.restype = C.c_char_p
        flags = L.apdgicp_build_flags().decode()
    except AttributeError as e:
        raise RuntimeError(f"{p} does not export apdgicp_build_flags (ABI < 6): rebuild it") from e
    variant = flags.split("| variant:", 1)[1].strip() if "| variant:" in flags else "?"
    if variant and os.environ.get("APDGICP_ALLOW_VARIANT_LIB", "0") != "1":
        raise RuntimeError(f"{p} is an experiment build ({variant}), not the product: rebuild it, or set APDGICP_ALLOW_VARIANT_LIB=1 for an A/B run")
    L.apdgicp_abi_version.restype = i32
    L.apdgicp_last_error.restype = C.c_char_p
    L.apdgicp_device_count.argtypes = [C.POINTER(i32)]
    L.apdgicp_default_params.argtypes = [C.POINTER(Params)]
    L.apdgicp_default_params.restype = None
    L.apdgicp_create.argtypes = [C.POINTER(Params), i32, vp, C.POINTER(vp)]
    L.apdgicp_destroy.argtypes = [vp]
    L.apdgicp_set_params.argtypes = [vp, C.POINTER(Params)]
    L.apdgicp_get_params.argtypes = [vp, C.POINTER(Params)]
    for f in (L.apdgicp_set_source, L.apdgicp_set_target):
        f.argtypes = [vp, vp, i64, i64, i32, u64]
    for f in (L.apdgicp_clear_source, L.apdgicp_clear_target, L.apdgicp_swap_source_and_target, L.apdgicp_synchronize):
        f.argtypes = [vp]
    L.apdgicp_compute_covariances.argtypes = [vp, i32]
    L.apdgicp_get_covariances.argtypes = [vp, i32, vp, i64]
    L.apdgicp_set_covariances.argtypes = [vp, i32, vp, i64]
    L.apdgicp_linearize.argtypes = [vp, vp, vp, vp, C.POINTER(dbl)]
    L.apdgicp_compute_error.argtypes = [vp, vp, C.POINTER(dbl)]
    L.apdgicp_get_correspondences.argtypes = [vp, vp, vp, i64]
    L.apdgicp_get_mahalanobis.argtypes = [vp, vp, i64]
    L.apdgicp_align.argtypes = [vp, vp, C.POINTER(Result)]
    L.apdgicp_align_host_loop.argtypes = [vp, vp, C.POINTER(Result)]
    L.apdgicp_get_final_hessian.argtypes = [vp, vp]
    L.apdgicp_set_trace.argtypes = [vp, i32]
    L.apdgicp_get_trace.argtypes = [vp, i64, vp, vp, vp, vp, C.POINTER(i64), i64, vp, C.POINTER(i64)]
    L.apdgicp_get_trace_step_norms.argtypes = [vp, i64, vp, C.POINTER(i64)]
    L.apdgicp_debug_atan2f.argtypes = [i32, vp, vp, vp, i64]
    L.apdgicp_nearest_neighbours.argtypes = [vp, vp, vp, vp, i64]
    L.apdgicp_nearest_neighbours_of.argtypes = [vp, vp, i64, i64, vp, vp]
    L.apdgicp_get_points.argtypes = [vp, i32, vp, i64]
    L.apdgicp_transform_source.argtypes = [vp, vp, vp, i64, i64]
    L.apdgicp_fitness_score.argtypes = [vp, vp, dbl, C.POINTER(dbl), C.POINTER(i64)]
    L.apdgicp_inlier_fraction.argtypes = [vp, vp, dbl, C.POINTER(dbl), C.POINTER(i64)]
    L.apdgicp_get_stream.argtypes = [vp, C.POINTER(vp)]
    L.apdgicp_batch_get_stream.argtypes = [vp, C.POINTER(vp)]
    L.apdgicp_wait_producer.argtypes = [vp, vp]
    L.apdgicp_batch_wait_producer.argtypes = [vp, vp]
    L.apdgicp_batch_create.argtypes = [C.POINTER(Params), i32, vp, C.POINTER(vp)]
    for f in (L.apdgicp_batch_destroy, L.apdgicp_batch_clear, L.apdgicp_batch_compute_covariances, L.apdgicp_batch_synchronize):
        f.argtypes = [vp]
    L.apdgicp_batch_set_params.argtypes = [vp, C.POINTER(Params)]
    L.apdgicp_batch_add_cloud.argtypes = [vp, vp, i64, i64, i32]
    L.apdgicp_batch_set_cloud.argtypes = [vp, i32, vp, i64, i64, i32]
    L.apdgicp_batch_set_clouds.argtypes = [vp, i32, i32, vp, vp, i64, i32]
    L.apdgicp_batch_align.argtypes = [vp, vp, i64, vp]
    L.apdgicp_batch_align_async.argtypes = [vp, vp, i64, C.POINTER(vp)]
    L.apdgicp_batch_align_enqueue.argtypes = [vp, vp, i64, C.POINTER(C.c_uint64)]
    L.apdgicp_batch_set_pair_groups.argtypes = [vp, C.c_int]
    L.apdgicp_batch_align_collect.argtypes = [vp, C.c_uint64, C.POINTER(vp), vp]
    L.apdgicp_batch_fitness.argtypes = [vp, vp, i64, vp, dbl, vp, vp]
    L.apdgicp_batch_copy_results.argtypes = [vp, vp, i64, i32]
    L.apdgicp_batch_set_profiling.argtypes = [vp, i32]
    L.apdgicp_batch_last_nn_time.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64)]
    L.apdgicp_batch_last_nn_profile.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64), C.POINTER(i64)]
    L.apdgicp_batch_last_ticks.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.apdgicp_batch_debug_stats.argtypes = [vp, vp]
    L.apdgicp_batch_debug_block_timeline.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.apdgicp_batch_pool_counters.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.apdgicp_batch_last_nn_kernel.argtypes = [vp, C.c_char_p, i32]
    L.apdgicp_submap_create.argtypes = [i32, vp, C.POINTER(vp)]
    L.apdgicp_submap_destroy.argtypes = [vp]
    L.apdgicp_submap_assemble.argtypes = [vp, i32, vp, vp, i64, i64, i32, vp, vp, C.POINTER(i64)]
    L.apdgicp_submap_points.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    L.apdgicp_submap_copy.argtypes = [vp, vp, i64, i32]
    if path is None:
        _lib = L
    return L


def source_stamp() -> str:
    """The fingerprint of the sources the loaded library was compiled from (apdgicp_source_stamp)."""
    return load_library().apdgicp_source_stamp().decode()


def build_flags() -> str:
    """The compiler flags of the loaded library and, behind "| variant:", its experiment defines (apdgicp_build_flags; empty = the product)."""
    return load_library().apdgicp_build_flags().decode()


def debug_atan2f(y, x, device: int = 0) -> np.ndarray:
    """atan2f as the kernels evaluate it on the device (include/apd_atan2f.h), for the bit-for-bit comparison with the host."""
    y = np.ascontiguousarray(y, dtype=np.float32).ravel()
    x = np.ascontiguousarray(x, dtype=np.float32).ravel()
    if y.shape != x.shape:
        raise ValueError("y and x must have the same number of elements")
    out = np.empty_like(y)
    _check(load_library().apdgicp_debug_atan2f(device, _ptr(y), _ptr(x), _ptr(out), y.size))
    return out


def default_params(**kw) -> Params:
    p = Params()
    load_library().apdgicp_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _check(rc: int) -> int:
    if rc < 0:
        raise ApdgicpError(rc, load_library().apdgicp_last_error().decode())
    return rc


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class DevicePoints:
    """n points in device memory, `stride_bytes` apart, xyz first (e.g. the cloud a SubmapAssembler holds);
    accepted wherever a cloud is: the pointer is passed straight through.  `owner` keeps the memory alive."""

    def __init__(self, ptr: int, n: int, stride_bytes: int, owner=None):
        self.ptr, self.n, self.stride_bytes, self.owner = int(ptr), int(n), int(stride_bytes), owner


def _cloud_arg(cloud):
    """-> (pointer, n, stride_bytes, on_device, keepalive).  Accepts numpy [n,>=3] float32, a
    torch CUDA/CPU float32 tensor of the same shape, or DevicePoints (device pointers are passed straight through)."""
    if isinstance(cloud, DevicePoints):
        return C.c_void_p(cloud.ptr), cloud.n, cloud.stride_bytes, 1, cloud
    if hasattr(cloud, "data_ptr"):  # torch tensor
        t = cloud
        if t.dim() != 2 or t.shape[1] < 3 or str(t.dtype) != "torch.float32" or t.stride(1) != 1:
            raise ValueError("cloud tensor must be [n, >=3] float32 with unit inner stride")
        return C.c_void_p(t.data_ptr()), t.shape[0], t.stride(0) * 4, 1 if t.is_cuda else 0, t
    a = np.asarray(cloud)
    if a.dtype != np.float32 or a.ndim != 2 or a.shape[1] < 3 or a.strides[1] != 4:
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.ndim != 2 or a.shape[1] < 3:
            raise ValueError("cloud must be [n, >=3]")
    return _ptr(a), a.shape[0], a.strides[0], 0, a


def _producer_stream(keep):
    """The stream a torch CUDA tensor was (by convention) produced on: torch's current stream.  None for anything else
    (raw DevicePoints: the caller orders them with wait_producer() itself)."""
    if hasattr(keep, "data_ptr") and getattr(keep, "is_cuda", False):
        import torch
        return C.c_void_p(torch.cuda.current_stream(keep.device).cuda_stream)
    return None


def _colmajor(T, dtype):
    return np.asfortranarray(np.asarray(T, dtype=dtype))


class FastAPDGICP:
    """One registration object (== one fast_gicp::FastAPDGICP) on one GPU."""

    def __init__(self, params: Params | None = None, device: int = 0, stream=None):
        self.L = load_library()
        self.params = params or default_params()
        self.h = C.c_void_p()
        _check(self.L.apdgicp_create(C.byref(self.params), device, stream, C.byref(self.h)))
        self.n_src = self.n_tgt = 0
        self.result = Result()
        self._converged = False
        self._final = np.eye(4, dtype=np.float32)
        self._keep = {}

    def close(self):
        if getattr(self, "h", None):
            self.L.apdgicp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the setters of the ROS factory (registrations.cpp:41-48)
    def _push(self):
        _check(self.L.apdgicp_set_params(self.h, C.byref(self.params)))

    def setNumThreads(self, n):  # A:34 -- no meaning on the GPU, accepted for drop-in compatibility
        pass

    def setTransformationEpsilon(self, eps):
        self.params.transformation_epsilon = eps
        self._push()

    def setRotationEpsilon(self, eps):
        self.params.rotation_epsilon = eps
        self._push()

    def setMaximumIterations(self, n):
        self.params.max_iterations = n
        self._push()

    def setMaxCorrespondenceDistance(self, d):
        self.params.max_correspondence_distance = d
        self._push()

    def setCorrespondenceRandomness(self, k):
        self.params.k_correspondences = k
        self._push()

    def setRegularizationMethod(self, m):
        self.params.regularization = m
        self._push()

    def setDistVar(self, v):
        self.params.distance_variance = v
        self._push()

    def setAzimuthVar(self, v):
        self.params.azimuth_variance_deg = v
        self._push()

    def setElevationVar(self, v):
        self.params.elevation_variance_deg = v
        self._push()

    def setTransformOrder(self, linear_chain: bool):
        """fp32 summation order of T * p (A:149): False = pairwise (Eigen >= 3.3, default), True = linear chain (Eigen 3.2); include/apdgicp_hip.h"""
        self.params.flags = (self.params.flags | FLAG_XF_LINEAR_CHAIN) if linear_chain else (self.params.flags & ~FLAG_XF_LINEAR_CHAIN)
        self._push()

    def setAlgebraicAPD(self, on: bool):
        """APDGICP_FLAG_ALGEBRAIC_APD (opt-in, include/apdgicp_hip.h): the sensor model of A:167-184 from coordinate ratios instead of fp32 angles"""
        self.params.flags = (self.params.flags | FLAG_ALGEBRAIC_APD) if on else (self.params.flags & ~FLAG_ALGEBRAIC_APD)
        self._push()

    def setInitialLambdaFactor(self, v):
        self.params.lm_init_lambda_factor = v
        self._push()

    def set_params(self, params: Params):
        self.params = params
        self._push()

    # ---- clouds (A:90-108); token = identity of the caller's cloud object (pointer equality in the reference)
    # Device-resident inputs (torch CUDA tensors): the handle's stream first waits for torch's current stream (the producer),
    # and the tensor is retained until the handle has certainly consumed it -- the next call that waits for the handle's
    # stream -- so a temporary (`g.setInputSource(x.cuda())`) cannot be recycled by torch's allocator under the pack kernel.
    def _set_cloud(self, fn, slot, cloud, token):
        p, n, stride, dev, keep = _cloud_arg(cloud)
        if dev:
            if self._keep.get(slot) is not None:   # an earlier device cloud of this slot may still be queued for packing
                _check(self.L.apdgicp_synchronize(self.h))
                self._keep.clear()
            ps = _producer_stream(keep)
            if ps is not None:
                _check(self.L.apdgicp_wait_producer(self.h, ps))
        _check(fn(self.h, p, n, stride, dev, token))
        if dev:
            self._keep[slot] = keep
        return n

    def wait_producer(self, stream_ptr: int = 0):
        """Orders the handle behind everything queued so far on the given hipStream_t (0: the legacy default stream)."""
        _check(self.L.apdgicp_wait_producer(self.h, C.c_void_p(stream_ptr)))

    def setInputSource(self, cloud, token: int = 0):
        self.n_src = self._set_cloud(self.L.apdgicp_set_source, SOURCE, cloud, token)

    def setInputTarget(self, cloud, token: int = 0):
        self.n_tgt = self._set_cloud(self.L.apdgicp_set_target, TARGET, cloud, token)

    def clearSource(self):
        _check(self.L.apdgicp_clear_source(self.h))
        self.n_src = 0

    def clearTarget(self):
        _check(self.L.apdgicp_clear_target(self.h))
        self.n_tgt = 0

    def swapSourceAndTarget(self):
        _check(self.L.apdgicp_swap_source_and_target(self.h))
        self.n_src, self.n_tgt = self.n_tgt, self.n_src

    # ---- covariances (H:64-74)
    def _n(self, which):
        return self.n_src if which == SOURCE else self.n_tgt

    def computeCovariances(self, which):
        _check(self.L.apdgicp_compute_covariances(self.h, which))

    def _get_covs(self, which):
        n = self._n(which)
        out = np.empty((n, 16))
        _check(self.L.apdgicp_get_covariances(self.h, which, _ptr(out), n))
        return out.reshape(n, 4, 4).transpose(0, 2, 1).copy()  # column-major -> numpy [n,4,4]

    def getSourceCovariances(self):
        return self._get_covs(SOURCE)

    def getTargetCovariances(self):
        return self._get_covs(TARGET)

    def _set_covs(self, which, covs):
        covs = np.asarray(covs, dtype=np.float64)
        n = self._n(which)
        if covs.shape == (n, 3, 3):
            full = np.zeros((n, 4, 4))
            full[:, :3, :3] = covs
            covs = full
        buf = np.ascontiguousarray(covs.transpose(0, 2, 1)).reshape(n, 16)
        _check(self.L.apdgicp_set_covariances(self.h, which, _ptr(buf), n))

    def setSourceCovariances(self, covs):
        self._set_covs(SOURCE, covs)

    def setTargetCovariances(self, covs):
        self._set_covs(TARGET, covs)

    # ---- probes (L:45-52, A:198-298)
    def linearize(self, T, want_Hb: bool = True):
        Tc = _colmajor(T, np.float64)
        H = np.zeros((6, 6), order="F")
        b = np.zeros(6)
        cost = C.c_double()
        _check(self.L.apdgicp_linearize(self.h, _ptr(Tc), _ptr(H) if want_Hb else None, _ptr(b) if want_Hb else None, C.byref(cost)))
        return cost.value, (np.ascontiguousarray(H) if want_Hb else None), (b if want_Hb else None)

    def evaluateCost(self, relative_pose, want_Hb: bool = False):  # L:50-52 (Matrix4f pose)
        return self.linearize(np.asarray(relative_pose, dtype=np.float32).astype(np.float64), want_Hb)

    def compute_error(self, T) -> float:
        Tc = _colmajor(T, np.float64)
        cost = C.c_double()
        _check(self.L.apdgicp_compute_error(self.h, _ptr(Tc), C.byref(cost)))
        return cost.value

    def correspondences(self):
        corr = np.empty(self.n_src, dtype=np.int32)
        sqd = np.empty(self.n_src, dtype=np.float32)
        _check(self.L.apdgicp_get_correspondences(self.h, _ptr(corr), _ptr(sqd), self.n_src))
        return corr, sqd

    def mahalanobis(self) -> np.ndarray:
        out = np.empty((self.n_src, 16))
        _check(self.L.apdgicp_get_mahalanobis(self.h, _ptr(out), self.n_src))
        return out.reshape(-1, 4, 4).transpose(0, 2, 1)[:, :3, :3].copy()

    # ---- align (pcl::Registration::align -> A:121-130 -> L:55-80)
    def align(self, guess=None, host_loop: bool = False, want_output: bool = False):
        g = None if guess is None else _colmajor(guess, np.float32)
        fn = self.L.apdgicp_align_host_loop if host_loop else self.L.apdgicp_align
        _check(fn(self.h, _ptr(g) if g is not None else None, C.byref(self.result)))
        self._keep.clear()   # align waits for the handle's stream: every queued pack has run
        self._converged = bool(self.result.converged)
        self._final = self.result.matrix()
        if want_output:
            return self.transformSource(self._final)
        return self._final

    def hasConverged(self) -> bool:
        return self._converged

    def getFinalTransformation(self) -> np.ndarray:
        return self._final

    def setTrace(self, enable: bool = True):
        """Debug: record the optimiser's per-iteration trace in the aligns that follow (include/apdgicp_hip.h: apdgicp_set_trace)."""
        _check(self.L.apdgicp_set_trace(self.h, 1 if enable else 0))

    def trace(self):
        """dict(lambda, rho, y0, yi, poses) of the last align: per LM trial the lambda the step was solved with, its rho and the two
        costs rho compares (L:137-146); per completed outer iteration the pose x0 behind it, [n, 4, 4] row-major."""
        cap_p = max(1, int(self.params.max_iterations))
        cap_t = cap_p * max(1, int(self.params.lm_max_iterations))
        lam, rho, y0, yi, poses = np.zeros(cap_t), np.zeros(cap_t), np.zeros(cap_t), np.zeros(cap_t), np.zeros((cap_p, 16))
        nt, npo = C.c_int64(), C.c_int64()
        _check(self.L.apdgicp_get_trace(self.h, cap_t, _ptr(lam), _ptr(rho), _ptr(y0), _ptr(yi), C.byref(nt), cap_p, _ptr(poses), C.byref(npo)))
        nt_, np_ = min(nt.value, cap_t), min(npo.value, cap_p)
        return {"lambda": lam[:nt_].copy(), "rho": rho[:nt_].copy(), "y0": y0[:nt_].copy(), "yi": yi[:nt_].copy(),
                "poses": poses[:np_].reshape(-1, 4, 4).transpose(0, 2, 1).copy()}

    def getFinalHessian(self) -> np.ndarray:
        H = np.zeros((6, 6), order="F")
        _check(self.L.apdgicp_get_final_hessian(self.h, _ptr(H)))
        return np.ascontiguousarray(H)

    def transformSource(self, T) -> np.ndarray:
        out = np.empty((self.n_src, 3), dtype=np.float32)
        Tc = _colmajor(T, np.float32)
        _check(self.L.apdgicp_transform_source(self.h, _ptr(Tc), _ptr(out), self.n_src, 12))
        return out

    def getFitnessScore(self, max_range: float = float(np.finfo(np.float64).max), T=None):
        Tc = _colmajor(self._final if T is None else T, np.float32)
        score, cnt = C.c_double(), C.c_int64()
        _check(self.L.apdgicp_fitness_score(self.h, _ptr(Tc), max_range, C.byref(score), C.byref(cnt)))
        self.last_inliers = cnt.value
        return score.value

    def nearestNeighbours(self, T=None):
        """(index, sq_dist) of the nearest target point of every T-transformed source point (T: the last pose by default): one
        batched device search, what pcl::search::KdTree::nearestKSearch(pt, 1, ...) returns point by point."""
        Tc = _colmajor(self._final if T is None else T, np.float32)
        idx = np.empty(self.n_src, dtype=np.int32)
        sqd = np.empty(self.n_src, dtype=np.float32)
        _check(self.L.apdgicp_nearest_neighbours(self.h, _ptr(Tc), _ptr(idx), _ptr(sqd), self.n_src))
        return idx, sqd

    def nearestNeighboursOf(self, queries):
        """(index, sq_dist) of the nearest target point of arbitrary host query points [n, >= 3] float32: one batched device pass
        (apdgicp_nearest_neighbours_of), what pcl::search::Search::nearestKSearch(cloud, indices, 1, ...) returns."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        idx = np.empty(len(q), dtype=np.int32)
        sqd = np.empty(len(q), dtype=np.float32)
        _check(self.L.apdgicp_nearest_neighbours_of(self.h, _ptr(q), len(q), q.strides[0], _ptr(idx), _ptr(sqd)))
        return idx, sqd

    def getPoints(self, which) -> np.ndarray:
        n = self._n(which)
        out = np.empty((n, 3), dtype=np.float32)
        _check(self.L.apdgicp_get_points(self.h, which, _ptr(out), n))
        return out

    def stream_ptr(self) -> int:
        st = C.c_void_p()
        _check(self.L.apdgicp_get_stream(self.h, C.byref(st)))
        return st.value or 0

    def inlierFraction(self, max_correspondence_dist: float = 0.5, T=None) -> float:
        """ScanMatchingStatus.inlier_fraction (scan_matching_odometry_nodelet.cpp:701-712): strict `<` on the squared distance."""
        Tc = _colmajor(self._final if T is None else T, np.float32)
        frac, cnt = C.c_double(), C.c_int64()
        _check(self.L.apdgicp_inlier_fraction(self.h, _ptr(Tc), max_correspondence_dist, C.byref(frac), C.byref(cnt)))
        self.last_inliers = cnt.value
        return frac.value

    @property
    def nr_iterations(self):
        return int(self.result.iterations)


class BatchAPDGICP:
    """Many independent registrations on one GPU (apdgicp_batch_*)."""

    def __init__(self, params: Params | None = None, device: int = 0, stream=None):
        self.L = load_library()
        self.params = params or default_params()
        self.b = C.c_void_p()
        _check(self.L.apdgicp_batch_create(C.byref(self.params), device, stream, C.byref(self.b)))
        self.n_clouds = 0
        # device-resident inputs retained until the batch has consumed them: [objects set since the last enqueue], and per
        # ticket the objects its pack kernels read (released by collect / synchronize / a blocking align)
        self._keep_new, self._keep_ticket = [], {}

    def close(self):
        if getattr(self, "b", None):
            self.L.apdgicp_batch_destroy(self.b)
            self.b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, params: Params):
        self.params = params
        _check(self.L.apdgicp_batch_set_params(self.b, C.byref(params)))

    def clear(self):
        _check(self.L.apdgicp_batch_clear(self.b))
        self.n_clouds = 0

    def _device_input(self, keep, producer_wait: bool = True):
        if producer_wait:
            ps = _producer_stream(keep[0] if isinstance(keep, list) and keep else keep)
            if ps is not None:
                _check(self.L.apdgicp_batch_wait_producer(self.b, ps))
        self._keep_new.append(keep)
        if len(self._keep_new) > 64:
            # A caller that sets clouds again and again without ever reaching a sync point (enqueue / align / synchronize) must not
            # pin tensors without bound -- but a tensor may only be let go once the pack kernel that reads it has run, or torch's
            # caching allocator hands its memory to somebody else under the kernel's feet.  So: wait for the handle, then drop all.
            _check(self.L.apdgicp_batch_synchronize(self.b))
            del self._keep_new[:-1]

    def wait_producer(self, stream_ptr: int = 0):
        """Orders the batch behind everything queued so far on the given hipStream_t (0: the legacy default stream)."""
        _check(self.L.apdgicp_batch_wait_producer(self.b, C.c_void_p(stream_ptr)))

    def add_cloud(self, cloud) -> int:
        p, n, stride, dev, keep = _cloud_arg(cloud)
        if dev:
            self._device_input(keep)
        idx = _check(self.L.apdgicp_batch_add_cloud(self.b, p, n, stride, dev))
        self.n_clouds = idx + 1
        return idx

    def set_cloud(self, index: int, cloud) -> int:
        p, n, stride, dev, keep = _cloud_arg(cloud)
        if dev:
            self._device_input(keep)
        idx = _check(self.L.apdgicp_batch_set_cloud(self.b, index, p, n, stride, dev))
        self.n_clouds = max(self.n_clouds, idx + 1)
        return idx

    @staticmethod
    def pack_clouds(clouds):
        """The C-side argument block of set_clouds (pointer array, sizes, stride, memory space) built once: a caller whose
        clouds live in persistent buffers -- like a C caller holding a pointer array -- passes it to set_clouds on every
        call instead of paying the per-tensor Python bookkeeping again.  Keeps the tensors alive."""
        args = [_cloud_arg(c) for c in clouds]
        stride, dev = args[0][2], args[0][3]
        if any(a[2] != stride or a[3] != dev for a in args):
            raise ValueError("set_clouds needs one stride and one memory space")
        ptrs = (C.c_void_p * len(args))(*[a[0] for a in args])
        ns = (C.c_int64 * len(args))(*[a[1] for a in args])
        return ("packed_clouds", ptrs, ns, stride, dev, len(args), [a[4] for a in args])

    def set_clouds(self, first_index: int, clouds, producer_wait: bool = True):
        """clouds: list of torch CUDA tensors (or numpy arrays), all with the same row stride -- or pack_clouds(list).
        producer_wait=False: the caller guarantees that the inputs are complete (long-lived resident buffers), no wait on
        torch's current stream is inserted -- in a multi-rank job that stream carries the RCCL all-gathers."""
        packed = clouds if isinstance(clouds, tuple) and clouds and clouds[0] == "packed_clouds" else self.pack_clouds(clouds)
        _, ptrs, ns, stride, dev, n, keep = packed
        if dev and keep:
            self._device_input(keep, producer_wait)   # (one producer wait for the whole list: torch's current stream)
        _check(self.L.apdgicp_batch_set_clouds(self.b, first_index, n, ptrs, ns, stride, dev))
        self.n_clouds = max(self.n_clouds, first_index + n)

    def compute_covariances(self):
        _check(self.L.apdgicp_batch_compute_covariances(self.b))
        self._keep_new.clear()   # the call waits for the stream (error flag): every queued pack has run

    @staticmethod
    def make_pairs(pairs, guesses=None):
        arr = (Pair * len(pairs))()
        for i, (s, t) in enumerate(pairs):
            arr[i].source_cloud, arr[i].target_cloud = int(s), int(t)
            g = np.eye(4, dtype=np.float32) if guesses is None else np.asarray(guesses[i], dtype=np.float32)
            arr[i].guess[:] = g.T.reshape(-1).tolist()
        return arr

    def align(self, pairs, guesses=None) -> np.ndarray:
        """pairs: list of (source_cloud, target_cloud) or a prepared Pair array.  Returns a
        structured numpy array (RESULT_DTYPE); T is column-major (use result_matrix)."""
        arr = pairs if isinstance(pairs, C.Array) else self.make_pairs(pairs, guesses)
        out = np.zeros(len(arr), dtype=RESULT_DTYPE)
        _check(self.L.apdgicp_batch_align(self.b, arr, len(arr), _ptr(out)))
        self._keep_new.clear(), self._keep_ticket.clear()   # a blocking align: every queued pack has run
        return out

    def align_async(self, pairs, guesses=None):
        """Runs the batch and leaves the results on the device; returns (device_pointer, n_bytes)."""
        arr = pairs if isinstance(pairs, C.Array) else self.make_pairs(pairs, guesses)
        dptr = C.c_void_p()
        _check(self.L.apdgicp_batch_align_async(self.b, arr, len(arr), C.byref(dptr)))
        self._keep_new.clear()   # returns behind its last status poll: every queued pack has run
        return dptr.value, len(arr) * RESULT_DTYPE.itemsize

    def set_pair_groups(self, max_groups: int):
        """At most this many pair groups (HIP streams) per batch; 1 for a handle that shares the GPU with other busy handles."""
        _check(self.L.apdgicp_batch_set_pair_groups(self.b, int(max_groups)))

    def align_enqueue(self, pairs, guesses=None) -> int:
        """Launches the whole batch and returns its ticket without waiting (Gauss-Newton; an LM batch is complete on return).
        The clouds of the NEXT batch may be set and enqueued before this one is collected (two batches in flight)."""
        arr = pairs if isinstance(pairs, C.Array) else self.make_pairs(pairs, guesses)
        ticket = C.c_uint64()
        _check(self.L.apdgicp_batch_align_enqueue(self.b, arr, len(arr), C.byref(ticket)))
        self._keep_ticket[ticket.value], self._keep_new = self._keep_new, []
        # (Gauss-Newton: the last two tickets are collectable; pooled LM batches: as many as the pool has lanes -- the library decides)
        self._ticket_pairs = {**{k: v for k, v in getattr(self, "_ticket_pairs", {}).items() if k + 32 >= ticket.value}, ticket.value: len(arr)}
        return ticket.value

    def align_collect(self, ticket: int, device: bool = False):
        """Waits for the batch of `ticket` (one of the last two enqueued).  Returns its records as a structured numpy array, or
        with device=True as a zero-copy torch uint8 CUDA tensor [n, 96] (valid until the second enqueue after its own)."""
        n = getattr(self, "_ticket_pairs", {}).get(ticket)
        if n is None:
            raise ValueError(f"ticket {ticket} is not one of the batches in flight")
        for t in [t for t in self._keep_ticket if t <= ticket]:   # stream order: everything up to this batch has run
            del self._keep_ticket[t]
        if device:
            dptr = C.c_void_p()
            _check(self.L.apdgicp_batch_align_collect(self.b, ticket, C.byref(dptr), None))
            return self._device_view(dptr.value, n * RESULT_DTYPE.itemsize)
        out = np.zeros(n, dtype=RESULT_DTYPE)
        _check(self.L.apdgicp_batch_align_collect(self.b, ticket, None, _ptr(out)))
        return out

    def _device_view(self, ptr, nbytes):
        import torch
        views = self.__dict__.setdefault("_views", {})
        if (ptr, nbytes) not in views:
            class _View:  # __cuda_array_interface__ is honoured by torch.as_tensor on ROCm builds as well
                pass
            v = _View()
            v.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
            if len(views) > 8:
                views.clear()
            views[(ptr, nbytes)] = torch.as_tensor(v, device="cuda").view(-1, RESULT_DTYPE.itemsize)
        return views[(ptr, nbytes)]

    def align_device(self, pairs, guesses=None):
        """Runs the batch and returns the result records as a zero-copy torch uint8 CUDA tensor [n, 96] over the engine's own
        result buffer (valid until the next align): what RCCL all-gathers, with no staging copy and no extra synchronisation
        (align_async returns after its last status poll, so the records are final)."""
        import torch
        ptr, nbytes = self.align_async(pairs, guesses)
        cached = getattr(self, "_result_view", None)
        if cached is not None and cached[0] == (ptr, nbytes):  # the engine's buffer does not move between equal batches
            return cached[1]

        class _View:  # __cuda_array_interface__ is honoured by torch.as_tensor on ROCm builds as well
            pass
        v = _View()
        v.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
        t = torch.as_tensor(v, device="cuda").view(-1, RESULT_DTYPE.itemsize)
        self._result_view = ((ptr, nbytes), t)
        return t

    def fitness(self, pairs, T=None, max_range: float = float(np.finfo(np.float64).max), guesses=None):
        """getFitnessScore(max_range) of every pair at poses T ([n,4,4] row-major numpy; None = the poses of the
        last align of the same pair list).  Returns (scores float64[n], inliers int64[n])."""
        arr = pairs if isinstance(pairs, C.Array) else self.make_pairs(pairs, guesses)
        n = len(arr)
        Tc = None
        if T is not None:
            Tc = np.ascontiguousarray(np.asarray(T, dtype=np.float32).reshape(n, 4, 4).transpose(0, 2, 1)).reshape(n, 16)
        scores = np.zeros(n, dtype=np.float64)
        inl = np.zeros(n, dtype=np.int64)
        _check(self.L.apdgicp_batch_fitness(self.b, arr, n, _ptr(Tc) if Tc is not None else None, max_range, _ptr(scores), _ptr(inl)))
        return scores, inl

    def stream_ptr(self) -> int:
        """The handle's hipStream_t as an integer (wrap it with torch.cuda.ExternalStream to record events on it)."""
        st = C.c_void_p()
        _check(self.L.apdgicp_batch_get_stream(self.b, C.byref(st)))
        return st.value or 0

    def synchronize(self):
        _check(self.L.apdgicp_batch_synchronize(self.b))
        self._keep_new.clear(), self._keep_ticket.clear()

    def copy_results_to(self, dst, n_pairs: int):
        """dst: torch uint8 tensor (device or host) or numpy array with room for n_pairs records."""
        if hasattr(dst, "data_ptr"):
            _check(self.L.apdgicp_batch_copy_results(self.b, C.c_void_p(dst.data_ptr()), n_pairs, 1 if dst.is_cuda else 0))
        else:
            _check(self.L.apdgicp_batch_copy_results(self.b, _ptr(dst), n_pairs, 0))

    def set_profiling(self, enable: bool):
        _check(self.L.apdgicp_batch_set_profiling(self.b, 1 if enable else 0))

    def last_nn_time(self):
        ms, n = C.c_double(), C.c_int64()
        _check(self.L.apdgicp_batch_last_nn_time(self.b, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def debug_stats(self):
        out = np.zeros(16, dtype=np.uint64)
        _check(self.L.apdgicp_batch_debug_stats(self.b, _ptr(out)))
        return out

    def debug_block_timeline(self, capacity: int = 8192) -> np.ndarray:
        """APDGICP_STATS=2: [n, 3] uint64 -- start, end (100 MHz ticks) and (pair << 32 | block index) of every block of the last dense search launch"""
        out = np.zeros((capacity, 3), dtype=np.uint64)
        n = C.c_int64()
        _check(self.L.apdgicp_batch_debug_block_timeline(self.b, _ptr(out), capacity, C.byref(n)))
        return out[:n.value]

    def pool_counters(self):
        """(chunks, ticks, slot_ticks) the handle's pair pool has enqueued so far (apdgicp_batch_pool_counters)"""
        a, b_, c = C.c_int64(), C.c_int64(), C.c_int64()
        _check(self.L.apdgicp_batch_pool_counters(self.b, C.byref(a), C.byref(b_), C.byref(c)))
        return a.value, b_.value, c.value

    def last_nn_profile(self):
        ms, n, pr = C.c_double(), C.c_int64(), C.c_int64()
        _check(self.L.apdgicp_batch_last_nn_profile(self.b, C.byref(ms), C.byref(n), C.byref(pr)))
        return ms.value, n.value, pr.value

    def last_nn_kernel(self) -> str:
        buf = C.create_string_buffer(96)
        _check(self.L.apdgicp_batch_last_nn_kernel(self.b, buf, 96))
        return buf.value.decode().strip("()")

    def last_ticks(self):
        a, s, t = C.c_int(), C.c_int(), C.c_int()
        _check(self.L.apdgicp_batch_last_ticks(self.b, C.byref(a), C.byref(s), C.byref(t)))
        return a.value, s.value, t.value


def result_matrix(rec) -> np.ndarray:
    return np.asarray(rec["T"], dtype=np.float32).reshape(4, 4).T.copy()
