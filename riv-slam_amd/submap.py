"""Scan-to-submap target assembly on the GPU: the step in front of registration_s2m->setInputTarget
in scan-to-map mode (/root/reference/radar_graph_slam/apps/scan_matching_odometry_nodelet.cpp:606-618) --
transform the clouds of the last `max_submap_frames` keyframes into the newest keyframe's frame,
concatenate, downsample() (:412-422, pcl::VoxelGrid with `downsample_resolution`,
preprocessing_nodelet.cpp:137-144) and make the result the registration target, without the
submap ever leaving the device.  Host side of include/apdgicp_hip.h's apdgicp_submap_* entry points.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .registration import DevicePoints, _check, _cloud_arg, _ptr, load_library

POINT_XYZI_INTENSITY_OFFSET = 16  # pcl::PointXYZI: {x, y, z, pad, intensity, pad[3]} -- 32 bytes per point


class SubmapAssembler:
    def __init__(self, device: int = 0, stream=None):
        self.L = load_library()
        self.h = C.c_void_p()
        _check(self.L.apdgicp_submap_create(device, C.c_void_p(stream) if stream else None, C.byref(self.h)))
        self.n = 0

    def __del__(self):
        try:
            if self.h:
                self.L.apdgicp_submap_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def assemble(self, clouds, rel_poses=None, leaf=None, intensity_column: int | None = 3) -> int:
        """clouds: list of [n, >=3] float32 arrays (numpy, or torch CUDA tensors -- all in the same memory space and with
        the same row stride); rel_poses: list of 4x4 (row-major numpy, double); leaf: float, 3 floats or None (no
        downsampling); intensity_column: column that holds the intensity (None or a column the clouds do not have: 0 is kept).
        Returns the number of points of the assembled cloud."""
        args = [_cloud_arg(c) for c in clouds]
        stride, dev = args[0][2], args[0][3]
        if any(a[2] != stride or a[3] != dev for a in args):
            raise ValueError("assemble needs one row stride and one memory space")
        ptrs = (C.c_void_p * len(args))(*[a[0] for a in args])
        ns = (C.c_int64 * len(args))(*[a[1] for a in args])
        ioff = -1
        if intensity_column is not None and 4 * (intensity_column + 1) <= stride:
            ioff = 4 * intensity_column
        poses = None
        if rel_poses is not None:
            if len(rel_poses) != len(args):
                raise ValueError("one pose per cloud")
            poses = np.ascontiguousarray(np.stack([np.asarray(T, dtype=np.float64).T.reshape(-1) for T in rel_poses]))
        lf = None
        if leaf is not None:
            lf = np.ascontiguousarray(np.broadcast_to(np.asarray(leaf, dtype=np.float32), (3,)))
        n_out = C.c_int64()
        _check(self.L.apdgicp_submap_assemble(self.h, len(args), ptrs, ns, stride, ioff, dev, _ptr(poses) if poses is not None else None,
                                              _ptr(lf) if lf is not None else None, C.byref(n_out)))
        self.n = n_out.value
        return self.n

    def points(self) -> DevicePoints:
        """the assembled cloud in device memory ({x, y, z, intensity}, 16-byte stride), valid until the next assemble"""
        p, n = C.c_void_p(), C.c_int64()
        _check(self.L.apdgicp_submap_points(self.h, C.byref(p), C.byref(n)))
        return DevicePoints(p.value or 0, n.value, 16, owner=self)

    def to_numpy(self) -> np.ndarray:
        out = np.empty((self.n, 4), dtype=np.float32)
        if self.n:
            _check(self.L.apdgicp_submap_copy(self.h, _ptr(out), self.n, 0))
        return out


def relative_poses(odoms, newest=None):
    """rel_pose_i = odom_i^-1 * odom_newest (scan_matching_odometry_nodelet.cpp:609), 4x4 doubles"""
    newest = np.asarray(odoms[-1] if newest is None else newest, dtype=np.float64)
    return [np.linalg.inv(np.asarray(o, dtype=np.float64)) @ newest for o in odoms]


def update_submap_target(registration, keyframe_clouds, keyframe_odoms, max_submap_frames: int, leaf, assembler: SubmapAssembler,
                         intensity_column: int | None = 3) -> int:
    """The `if (enable_scan_to_map)` block at :606-618.  keyframe_clouds / keyframe_odoms include the keyframe that was
    just pushed (keyframes.back()); like the reference loop, the submap is built from the keyframes BEFORE it:
    i in [max(0, size - max_submap_frames), size - 1).  Sets the result as `registration`'s target and returns its size
    (0: fewer than two keyframes, the reference then hands PCL an empty cloud; nothing is set here)."""
    size = len(keyframe_clouds)
    first = max(0, size - max_submap_frames)
    idx = list(range(first, size - 1))
    if not idx:
        return 0
    poses = relative_poses([keyframe_odoms[i] for i in idx], keyframe_odoms[-1])
    n = assembler.assemble([keyframe_clouds[i] for i in idx], poses, leaf, intensity_column)
    if n:
        registration.setInputTarget(assembler.points())
    return n
