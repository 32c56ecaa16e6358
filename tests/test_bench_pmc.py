"""bench.py's PMC-derived fields (roofline.traffic, roofline_issue, roofline_issue_step) come from a committed profile of a separate
rocprofv3 --pmc run.  They may only be reported when that profile was collected from the kernel sources the bench runs: the
file is stamped with a hash of riv-slam_amd/csrc/* and bench.load_pmc refuses any other stamp (CPU only, no GPU needed)."""
import importlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

build = importlib.import_module("riv-slam_amd.build")


def test_stamp_follows_the_kernel_sources(tmp_path, monkeypatch):
    a = build.source_stamp()
    assert len(a) == 16 and a == build.source_stamp()
    csrc = tmp_path / "csrc"
    csrc.mkdir()
    for f in os.listdir(build.CSRC):
        if f.endswith((".hip", ".hpp")):
            (csrc / f).write_bytes(open(os.path.join(build.CSRC, f), "rb").read())
    monkeypatch.setattr(build, "CSRC", str(csrc))
    assert build.source_stamp() == a          # same bytes, same stamp
    with open(csrc / "apd_kernels.hpp", "ab") as fh:
        fh.write(b"\n// one more line\n")
    assert build.source_stamp() != a          # any change of a kernel source changes it


def test_a_profile_of_other_sources_is_refused(tmp_path):
    stamp = build.source_stamp()
    good, stale, bare = tmp_path / "good.json", tmp_path / "stale.json", tmp_path / "bare.json"
    good.write_text(json.dumps({"source_stamp": stamp, "SQ_INSTS_VALU": 1.0}))
    stale.write_text(json.dumps({"source_stamp": "0123456789abcdef", "SQ_INSTS_VALU": 1.0}))
    bare.write_text(json.dumps({"SQ_INSTS_VALU": 1.0}))   # a file from before the stamp existed (round 2's)
    pmc, why = bench.load_pmc(str(good))
    assert pmc and why is None and pmc["SQ_INSTS_VALU"] == 1.0
    for path in (stale, bare):
        pmc, why = bench.load_pmc(str(path))
        assert pmc is None and "refresh_evidence" in why and stamp in why
    pmc, why = bench.load_pmc(str(tmp_path / "missing.json"))
    assert pmc is None and why


def test_the_committed_profile_is_either_current_or_refused():
    """Never silently stale: the committed file is used by bench.py only when its stamp is that of the sources in the tree."""
    path = os.path.join(ROOT, "profiles", "pmc_nn_latest.json")
    pmc, why = bench.load_pmc(path)
    have = json.load(open(path)).get("source_stamp")
    if have == build.source_stamp():
        assert pmc is not None and why is None
    else:
        assert pmc is None and why
        pytest.xfail(f"profiles/pmc_nn_latest.json is stale ({have} vs {build.source_stamp()}): bench.py prints null for the PMC fields until "
                     "tools/refresh_evidence.sh has been re-run on the GPU box and the file committed")


def test_a_library_built_from_other_sources_is_refused(tmp_path, monkeypatch):
    """The library carries the stamp of the sources it was compiled from (apdgicp_source_stamp); the loader compares it with
    the sources on disk and refuses a mismatch -- age plays no part (a prebuilt library pushed with a tree of newer mtimes used
    to pass the freshness test).  Simulated without a second compile: the sources "on disk" are a touched copy."""
    reg = importlib.import_module("riv-slam_amd.registration")
    import __graft_entry__ as g
    g.build()
    assert build.library_stamp() == build.source_stamp() and not build.needs_build()
    assert reg.load_library().apdgicp_source_stamp().decode() == build.source_stamp()
    csrc = tmp_path / "csrc"
    csrc.mkdir()
    for f in os.listdir(build.CSRC):
        if f.endswith((".hip", ".hpp")):
            (csrc / f).write_bytes(open(os.path.join(build.CSRC, f), "rb").read())
    with open(csrc / "apd_kernels.hpp", "ab") as fh:
        fh.write(b"\n// touched\n")
    monkeypatch.setattr(build, "CSRC", str(csrc))
    assert build.needs_build()                       # identity, not mtime: the library on disk is older AND of other sources
    os.utime(build.LIB)                               # ... and making it the newest file changes nothing
    assert build.needs_build()
    monkeypatch.setattr(reg, "_lib", None)
    with pytest.raises(RuntimeError, match="compiled from other sources"):
        reg.load_library()
    monkeypatch.setenv("APDGICP_ALLOW_STALE_LIB", "1")
    assert reg.load_library() is not None
