"""C++ multi-device path: ShardedBatchAlignerHip (one process, one host thread per GPU, ncclAllGather of the records) and
LoopVerifierHip (the C++ candidate selection of LoopDetector::matching).  Compiled and linked here (hipcc cross-links
amdhip64 + rccl without a GPU); run on the GPU box with the devices it has (one in the test pool)."""
import importlib
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "test_multi_device")


def build_exe():
    import __graft_entry__ as g
    g.build()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    lib_dir = os.path.join(ROOT, "riv-slam_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "riv-slam_amd", "cpp"), os.path.join(ROOT, "tests", "cpp", "test_multi_device.cpp"),
           "-L", lib_dir, "-lapdgicp_hip", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lrccl", "-o", EXE]
    subprocess.check_call(cmd)
    return EXE


def build_bench():
    import __graft_entry__ as g
    g.build()
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "bench_sharded")
    lib_dir = os.path.join(ROOT, "riv-slam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "riv-slam_amd", "cpp"), os.path.join(ROOT, "tests", "cpp", "bench_sharded.cpp"),
                           "-L", lib_dir, "-lapdgicp_hip", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lrccl", "-o", exe])
    return exe


def write_batch_file(path, clouds, pairs, guesses):
    """the file format of tests/cpp/test_multi_device.cpp and bench_sharded.cpp"""
    with open(path, "wb") as f:
        np.array([len(clouds)], dtype=np.int32).tofile(f)
        for c in clouds:
            np.array([len(c)], dtype=np.int32).tofile(f)
            np.ascontiguousarray(c[:, :3], dtype=np.float32).tofile(f)
        np.array([len(pairs)], dtype=np.int32).tofile(f)
        for (s_, t_), g in zip(pairs, guesses):
            np.array([s_, t_], dtype=np.int32).tofile(f)
            np.asfortranarray(g).T.astype(np.float32).tofile(f)


def test_multi_device_harness_compiles_and_links():
    exe = build_exe()
    assert subprocess.run([build_bench()], capture_output=True, text=True, timeout=120).stdout.startswith("compile-only")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "compile-only" in out.stdout


@pytest.mark.gpu
def test_sharded_cpp_equals_single_handle_and_python_verifier(scene, pkg, tmp_path):
    exe = build_exe()
    reg = importlib.import_module("riv-slam_amd.registration")
    lv = importlib.import_module("riv-slam_amd.loop_verifier")
    # one new keyframe (cloud 0) with 5 candidates + 2 unrelated pairs: 7 pairs, ragged sizes
    clouds, pairs, guesses = [], [], []
    tgt = None
    for i in range(5):
        s, t, _, g = scene.make_pair(1500 + 200 * i, 2048, scene.pair_seed(31, 0), "loop" if i % 2 else "odometry")
        if tgt is None:
            tgt = t
            clouds.append(tgt)
        rng = np.random.default_rng(100 + i)
        clouds.append((s + rng.normal(scale=0.02 * i, size=s.shape)).astype(np.float32))   # candidates of decreasing quality
        pairs.append((len(clouds) - 1, 0))
        guesses.append(g)
    for i in range(2):
        s, t, _, g = scene.make_pair(1024, 1300, scene.pair_seed(32, i), "odometry")
        clouds += [s, t]
        pairs.append((len(clouds) - 2, len(clouds) - 1))
        guesses.append(g)
    path = tmp_path / "batch.bin"
    with open(path, "wb") as f:
        np.array([len(clouds)], dtype=np.int32).tofile(f)
        for c in clouds:
            np.array([len(c)], dtype=np.int32).tofile(f)
            np.ascontiguousarray(c[:, :3], dtype=np.float32).tofile(f)
        np.array([len(pairs)], dtype=np.int32).tofile(f)
        for (s_, t_), g in zip(pairs, guesses):
            np.array([s_, t_], dtype=np.int32).tofile(f)
            np.asfortranarray(g).T.astype(np.float32).tofile(f)
    out = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("world ")]   # (RCCL may print its own lines)
    assert len(lines) == 1, out.stdout + out.stderr
    tok = lines[0].split()
    vals = dict(zip(tok[0::2], tok[1::2]))
    assert int(vals["world"]) >= 1 and int(vals["pairs"]) == 7
    assert vals["sharded_equals_single"] == "1" and vals["gathered_on_all_ranks"] == "1"
    assert vals["pipelined_equals_single"] == "1" and vals["errors_ok"] == "1"
    # the C++ selection == the Python mirror of LoopDetector::matching on the same candidates
    kw = dict(max_correspondence_distance=2.0, transformation_epsilon=0.01, azimuth_variance_deg=1.0)
    b = reg.BatchAPDGICP(reg.default_params(**kw))
    loop, scores, _ = lv.verify_candidates(b, clouds[0], clouds[1:6], guesses[:5], fitness_score_max_range=4.0, fitness_score_thresh=0.5)
    assert int(vals["candidates"]) == 5
    assert int(vals["loop_best"]) == (loop.candidate if loop else -1)
    if loop:
        assert float(vals["loop_score"]) == loop.fitness_score
    # InformationMatrixCalculatorHip (C++) == the Python mirror on pair 0 at the registered pose
    im = importlib.import_module("riv-slam_amd.information_matrix")
    single = reg.BatchAPDGICP(reg.default_params(**kw))
    single.set_clouds(0, clouds)
    T0 = reg.result_matrix(single.align(pairs, guesses)[0])
    calc = im.InformationMatrixCalculator()
    fs = calc.calc_fitness_score(clouds[pairs[0][1]], clouds[pairs[0][0]], T0)
    inf = im.information_from_fitness(calc.params, fs)
    assert float(vals["fitness"]) == fs and float(vals["inf00"]) == inf[0, 0] and float(vals["inf33"]) == inf[3, 3]


@pytest.mark.gpu
@pytest.mark.parametrize("optimizer", ("gn", "lm"))
def test_cpp_pipelined_bench_records_equal_the_python_path(scene, tmp_path, optimizer):
    """tests/cpp/bench_sharded.cpp (the C++ caller's form of bench.py's step: device-resident clouds re-registered every batch,
    several batches in flight per device, one all-gather per batch): its records must be those of a Python batch handle."""
    exe = build_bench()
    reg = importlib.import_module("riv-slam_amd.registration")
    P, n = 8, 3000
    clouds, pairs, guesses = [], [], []
    for p in range(P):
        s, t, _, g = scene.make_pair(n, n, scene.pair_seed(2, p), "odometry" if optimizer == "gn" else "loop")
        clouds += [s, t]
        pairs.append((2 * p, 2 * p + 1))
        guesses.append(g if optimizer == "gn" else np.eye(4, dtype=np.float32))
    path, out_path = tmp_path / "bench.bin", tmp_path / "records.bin"
    write_batch_file(path, clouds, pairs, guesses)
    out = subprocess.run([exe, str(path), optimizer, "6", "2", str(out_path)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stdout + out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    import json
    rep = json.loads(line)
    assert rep["records_stable"] == 1 and rep["pairs_per_device"] == P
    kw = (dict(optimizer=1, max_iterations=20, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0, azimuth_variance_deg=1.0)
          if optimizer == "gn" else dict(max_correspondence_distance=2.0, transformation_epsilon=0.1, azimuth_variance_deg=1.0))
    b = reg.BatchAPDGICP(reg.default_params(**kw))
    b.set_clouds(0, clouds)
    want = b.align(pairs, guesses)
    assert open(out_path, "rb").read() == want.tobytes()
