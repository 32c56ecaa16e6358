"""The C++ drop-in class (riv-slam_amd/cpp/fast_apdgicp_hip.hpp) behind pcl::Registration.
PCL is not installed here, so it is compiled against tests/pcl_shim (test-only stand-in)."""
import importlib
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "test_adapter")


def build_exe():
    import __graft_entry__ as g
    g.build()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    lib_dir = os.path.join(ROOT, "riv-slam_amd")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "tests", "pcl_shim"), "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "riv-slam_amd", "cpp"), os.path.join(ROOT, "tests", "cpp", "test_adapter.cpp"),
           "-L", lib_dir, "-lapdgicp_hip", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", EXE]
    subprocess.check_call(cmd)
    return EXE


def test_adapter_compiles_and_links():
    exe = build_exe()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "compile-only" in out.stdout


@pytest.mark.gpu
def test_adapter_matches_python_binding(golden, scene, tmp_path):
    exe = build_exe()
    reg = importlib.import_module("riv-slam_amd.registration")
    src, tgt, guess = golden["lin_source"], golden["lin_target"], golden["lin_guess"]
    path = tmp_path / "pair.bin"
    with open(path, "wb") as f:
        np.array([len(src), len(tgt)], dtype=np.int32).tofile(f)
        np.asfortranarray(guess).T.astype(np.float32).tofile(f)   # column-major
        src.astype(np.float32).tofile(f)
        tgt.astype(np.float32).tofile(f)
    out = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    vals = out.stdout.split()
    conv, iters = int(vals[0]), int(vals[1])
    T = np.array(vals[2:18], dtype=np.float32).reshape(4, 4).T
    p0 = np.array(vals[18:22], dtype=np.float32)
    kw = dict(max_correspondence_distance=2.0, transformation_epsilon=0.1, azimuth_variance_deg=1.0)
    g = reg.FastAPDGICP(reg.default_params(**kw))
    g.setInputSource(src)
    g.setInputTarget(tgt)
    Tp = g.align(guess)
    assert conv == int(g.hasConverged()) and iters == g.nr_iterations
    assert np.array_equal(T, Tp)
    assert list(golden["lm_launch_info"][:2]) == [conv, iters]
    want = Tp[:3, :3] @ src[0] + Tp[:3, 3]
    assert np.abs(p0[:3] - want).max() < 1e-4 and p0[3] == 42.0   # intensity carried over like pcl::transformPointCloud
    assert int(vals[22]) == 1   # setInputTargetDevice (device-resident submap as target): the same registration


@pytest.mark.gpu
def test_adapter_protocol_timing_mode(scene, tmp_path):
    """align.cpp's protocols and the odometry frame loop through the adapter with host clouds (tests/measure/odometry_protocol.py
    runs the same binary at 8192 points for profiles/)."""
    import json
    exe = build_exe()
    src, tgt, _, guess = scene.make_pair(2048, 2048, scene.pair_seed(2, 0), "odometry")
    path = tmp_path / "pair.bin"
    with open(path, "wb") as f:
        np.array([len(src), len(tgt)], dtype=np.int32).tofile(f)
        np.asfortranarray(guess).T.astype(np.float32).tofile(f)
        src.astype(np.float32).tofile(f)
        tgt.astype(np.float32).tofile(f)
    out = subprocess.run([exe, str(path), "--protocol"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["converged"] == 1 and 0.0 < d["inlier_fraction"] <= 1.0
    for k in ("align_cpp_single_ms", "align_cpp_100_times_per_call_ms", "align_cpp_100_times_reuse_per_call_ms", "odometry_frame_ms"):
        assert 0.0 < d[k]["p10"] <= d[k]["median"] <= d[k]["p90"] < 50.0, (k, d[k])
