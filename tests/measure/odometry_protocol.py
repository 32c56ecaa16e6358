#!/usr/bin/env python3
"""The real caller path, timed: host pcl::PointXYZI clouds through the C++ pcl::Registration adapter
(tests/cpp/test_adapter.cpp --protocol): fast_apdgicp/src/align.cpp's single / 100x / 100x-reuse protocols and
scan-to-keyframe odometry frames (scan_matching_odometry_nodelet.cpp:449-471), LM with the launch parameters.
usage (inside gpurun): python tests/measure/odometry_protocol.py [points] > profiles/rNN_odometry_protocol.json
ODOMETRY_KEEP_INPUT=dir keeps dir/pair.bin; then `rocprofv3 --kernel-trace -d out -- build/test_adapter dir/pair.bin --protocol` and
tools/rocpd_timeline.py show where a frame's time goes."""
import importlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_cpp_adapter import build_exe  # noqa: E402

scene = importlib.import_module("riv-slam_amd.scene")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
src, tgt, _, guess = scene.make_pair(n, n, scene.pair_seed(2, 0), "odometry")
exe = build_exe()
keep = os.environ.get("ODOMETRY_KEEP_INPUT")   # a directory: the input file stays there (for a profiler run of the same executable)
if keep:
    os.makedirs(keep, exist_ok=True)
with tempfile.TemporaryDirectory() as d:
    d = keep or d
    path = os.path.join(d, "pair.bin")
    with open(path, "wb") as f:
        np.array([len(src), len(tgt)], dtype=np.int32).tofile(f)
        np.asfortranarray(guess).T.astype(np.float32).tofile(f)   # column-major
        src.astype(np.float32).tofile(f)
        tgt.astype(np.float32).tofile(f)
    out = subprocess.run([exe, path, "--protocol"], capture_output=True, text=True, timeout=600)
if out.returncode != 0:
    sys.exit(out.stderr)
d = json.loads(out.stdout.strip().splitlines()[-1])
d["what"] = ("FastAPDGICPHip<PointXYZI> behind pcl::Registration (PCL shim), host clouds, LM with the launch parameters; wall clock around "
             "setInputTarget/setInputSource/align (+ the aligned output cloud)")
print(json.dumps(d))
