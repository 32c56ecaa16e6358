"""Seeded synthetic "radar-like structured scene" generator.

This is measurement/test infrastructure shared by bench.py and tests/: it only
produces input clouds, it computes nothing on the APD-GICP path.

Scene (world == target sensor frame): ground plane z=-1.5 m, 6 vertical wall
patches, 24 box clusters inside x in [2, 100] m.  Points are sampled on those
surfaces, kept when they fall inside the radar frustum (azimuth +-56.5 deg,
elevation +-15 deg, range 2..100 m; reference launch file
radar_graph_slam/launch/radar_graph_slam.launch:51-52,143) and perturbed with
polar measurement noise sigma_r = 0.00215*r (= dist_var/400,
fast_apdgicp_impl.hpp:169), sigma_az = sigma_el = 0.2 deg.  Source and target
are sampled independently (no point-to-point identity).
"""
from __future__ import annotations

import numpy as np

BASE_SEED = 20241022  # reference snapshot date; seed = BASE + 1000*config_id + pair_index

AZ_MAX = np.deg2rad(56.5)
EL_MAX = np.deg2rad(15.0)
R_MIN, R_MAX = 2.0, 100.0
SIGMA_R_FACTOR = 0.00215
SIGMA_ANG = np.deg2rad(0.2)


def pair_seed(config_id: int, pair_index: int = 0) -> int:
    return BASE_SEED + 1000 * int(config_id) + int(pair_index)


def rot_zyx(yaw: float, pitch: float, roll: float) -> np.ndarray:
    cy, sy = np.cos(yaw), np.sin(yaw)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cr, sr = np.cos(roll), np.sin(roll)
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return rz @ ry @ rx


def make_transform(t, yaw=0.0, pitch=0.0, roll=0.0) -> np.ndarray:
    T = np.eye(4)
    T[:3, :3] = rot_zyx(yaw, pitch, roll)
    T[:3, 3] = np.asarray(t, dtype=np.float64)
    return T


class Scene:
    """A set of planar rectangles (origin o, edge vectors u, v) with sampling weights."""

    def __init__(self, rng: np.random.Generator):
        rects = []  # (o, u, v, weight)
        # ground: one big rectangle, weight fixed (radar returns from ground are sparse)
        rects.append((np.array([0.0, -160.0, -1.5]), np.array([110.0, 0, 0]), np.array([0, 320.0, 0]), 0.30))
        # 6 wall patches: random yaw, 10-40 m long, 3 m high, standing on the ground
        for _ in range(6):
            L = rng.uniform(10.0, 40.0)
            yaw = rng.uniform(-np.pi, np.pi)
            cx = rng.uniform(10.0, 90.0)
            bearing = rng.uniform(-0.8, 0.8)
            c = np.array([cx * np.cos(bearing), cx * np.sin(bearing), -1.5])
            u = np.array([np.cos(yaw), np.sin(yaw), 0.0]) * L
            v = np.array([0, 0, 3.0])
            rects.append((c - 0.5 * u, u, v, 0.40 / 6))
        # 24 boxes (0.5-3 m), five faces each (bottom omitted)
        for _ in range(24):
            s = rng.uniform(0.5, 3.0, size=3)
            rng_c = rng.uniform(4.0, 95.0)
            bearing = rng.uniform(-0.9, 0.9)
            c = np.array([rng_c * np.cos(bearing), rng_c * np.sin(bearing), -1.5 + rng.uniform(0.0, 2.0)])
            yaw = rng.uniform(-np.pi, np.pi)
            ex = np.array([np.cos(yaw), np.sin(yaw), 0.0]) * s[0]
            ey = np.array([-np.sin(yaw), np.cos(yaw), 0.0]) * s[1]
            ez = np.array([0, 0, s[2]])
            o = c - 0.5 * ex - 0.5 * ey
            faces = [(o, ex, ez), (o + ey, ex, ez), (o, ey, ez), (o + ex, ey, ez), (o + ez, ex, ey)]
            for (fo, fu, fv) in faces:
                rects.append((fo, fu, fv, 0.30 / (24 * 5)))
        self.o = np.stack([r[0] for r in rects])
        self.u = np.stack([r[1] for r in rects])
        self.v = np.stack([r[2] for r in rects])
        w = np.array([r[3] for r in rects])
        self.w = w / w.sum()

    def sample(self, rng: np.random.Generator, n: int) -> np.ndarray:
        k = rng.choice(len(self.w), size=n, p=self.w)
        a = rng.uniform(size=(n, 1))
        b = rng.uniform(size=(n, 1))
        # ground samples: concentrate near the sensor like real returns (r ~ uniform in range, not area)
        return self.o[k] + a * self.u[k] + b * self.v[k]


def _in_frustum(p: np.ndarray) -> np.ndarray:
    r = np.linalg.norm(p, axis=1)
    az = np.arctan2(p[:, 1], p[:, 0])
    el = np.arctan2(p[:, 2], np.hypot(p[:, 0], p[:, 1]))
    return (r >= R_MIN) & (r <= R_MAX) & (np.abs(az) <= AZ_MAX) & (np.abs(el) <= EL_MAX)


def _polar_noise(rng: np.random.Generator, p: np.ndarray) -> np.ndarray:
    r = np.linalg.norm(p, axis=1)
    az = np.arctan2(p[:, 1], p[:, 0])
    el = np.arctan2(p[:, 2], np.hypot(p[:, 0], p[:, 1]))
    r = r + rng.normal(size=r.shape) * SIGMA_R_FACTOR * r
    az = az + rng.normal(size=r.shape) * SIGMA_ANG
    el = el + rng.normal(size=r.shape) * SIGMA_ANG
    ce = np.cos(el)
    return np.stack([r * ce * np.cos(az), r * ce * np.sin(az), r * np.sin(el)], axis=1)


def _observe(rng: np.random.Generator, scene: Scene, T_world_from_sensor: np.ndarray, n: int) -> np.ndarray:
    """Exactly n noisy points of the scene as seen by a sensor at the given pose."""
    Tinv = np.linalg.inv(T_world_from_sensor)
    out = np.empty((0, 3))
    while out.shape[0] < n:
        pw = scene.sample(rng, max(4 * n, 4096))
        ps = pw @ Tinv[:3, :3].T + Tinv[:3, 3]
        ps = ps[_in_frustum(ps)]
        out = np.concatenate([out, ps], axis=0)
    out = out[:n]
    return _polar_noise(rng, out).astype(np.float32)


def make_pair(n_src: int, n_tgt: int, seed: int, kind: str = "odometry"):
    """Return (source[N,3] f32, target[M,3] f32, T_true[4,4] f64 source->target, guess[4,4] f32).

    kind="odometry": t=(U[0.2,1.0], U[-0.1,0.1], U[-0.03,0.03]) m, yaw U[-2,2] deg,
      pitch/roll U[-0.3,0.3] deg; guess = truth perturbed by (5 cm, 0.5 deg).
    kind="loop": t up to 3 m, yaw up to 20 deg; guess = identity (loop_detector.cpp:225).
    """
    rng = np.random.default_rng(seed)
    scene = Scene(rng)
    if kind == "odometry":
        t = np.array([rng.uniform(0.2, 1.0), rng.uniform(-0.1, 0.1), rng.uniform(-0.03, 0.03)])
        yaw = np.deg2rad(rng.uniform(-2, 2))
        pitch, roll = np.deg2rad(rng.uniform(-0.3, 0.3, size=2))
    elif kind == "loop":
        d = rng.uniform(0.5, 3.0)
        ang = rng.uniform(-np.pi, np.pi)
        t = np.array([d * np.cos(ang), d * np.sin(ang), rng.uniform(-0.05, 0.05)])
        yaw = np.deg2rad(rng.uniform(-20, 20))
        pitch, roll = np.deg2rad(rng.uniform(-0.5, 0.5, size=2))
    else:
        raise ValueError(kind)
    T_true = make_transform(t, yaw, pitch, roll)  # source sensor pose in target frame
    target = _observe(rng, scene, np.eye(4), n_tgt)
    source = _observe(rng, scene, T_true, n_src)
    if kind == "odometry":
        dt = rng.normal(size=3)
        dt = 0.05 * dt / np.linalg.norm(dt)
        dyaw = np.deg2rad(0.5) * rng.choice([-1.0, 1.0])
        guess = make_transform(dt, dyaw, 0.0, 0.0) @ T_true
    else:
        guess = np.eye(4)
    return source, target, T_true, guess.astype(np.float32)


def make_keyframe_set(n_src: int, n_tgt: int, n_keyframes: int, seed: int):
    """One scan against the last `n_keyframes` keyframes of the SAME street (BASELINE configs[2], the candidates of
    scan_matching_odometry_nodelet.cpp:606-618 before they are merged into a submap): the sensor advances by an odometry
    step per keyframe, the scan is taken one step past the newest keyframe.
    Returns (source[N,3] f32, [target_k[M,3] f32], [T_true_k 4x4 f64 source->keyframe k], [guess_k 4x4 f32])."""
    rng = np.random.default_rng(seed)
    scene = Scene(rng)
    poses, T = [], np.eye(4)
    for _ in range(n_keyframes + 1):
        poses.append(T.copy())
        step = make_transform(np.array([rng.uniform(0.2, 1.0), rng.uniform(-0.1, 0.1), rng.uniform(-0.03, 0.03)]),
                              np.deg2rad(rng.uniform(-2, 2)), *np.deg2rad(rng.uniform(-0.3, 0.3, size=2)))
        T = T @ step
    targets = [_observe(rng, scene, P, n_tgt) for P in poses[:-1]]
    source = _observe(rng, scene, poses[-1], n_src)
    truths, guesses = [], []
    for P in poses[:-1]:
        Tk = np.linalg.inv(P) @ poses[-1]
        dt = rng.normal(size=3)
        dt = 0.05 * dt / np.linalg.norm(dt)
        truths.append(Tk)
        guesses.append((make_transform(dt, np.deg2rad(0.5) * rng.choice([-1.0, 1.0]), 0.0, 0.0) @ Tk).astype(np.float32))
    return source, targets, truths, guesses


def pose_error(T_ref: np.ndarray, T_est: np.ndarray):
    """(t_err [m], r_err [rad]) of delta = T_ref^-1 * T_est; metric of
    fast_apdgicp/src/test/gicp_test.cpp:73-78."""
    d = np.linalg.inv(np.asarray(T_ref, dtype=np.float64)) @ np.asarray(T_est, dtype=np.float64)
    t_err = float(np.linalg.norm(d[:3, 3]))
    c = (np.trace(d[:3, :3]) - 1.0) * 0.5
    r_err = float(np.arccos(np.clip(c, -1.0, 1.0)))
    if r_err < 1e-6:  # arccos loses precision near 0: use the skew part
        w = np.array([d[2, 1] - d[1, 2], d[0, 2] - d[2, 0], d[1, 0] - d[0, 1]]) * 0.5
        r_err = float(np.linalg.norm(w))
    return t_err, r_err
