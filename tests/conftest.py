import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def pytest_collection_modifyitems(config, items):
    """A hang (a rendezvous that never completes, a GPU that stops answering) fails the test that hangs instead of holding the run:
    every GPU test gets a ten-minute ceiling when pytest-timeout is installed (it is in this image)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "apdgicp_golden.npz")
    return dict(np.load(path))


@pytest.fixture(scope="session")
def scene():
    return importlib.import_module("riv-slam_amd.scene")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("riv-slam_amd")


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
