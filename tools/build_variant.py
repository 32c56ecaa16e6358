#!/usr/bin/env python3
"""Builds riv-slam_amd/_<name>.bin: the library with extra compiler flags (e.g. -DAPD_OCML_ATAN2F), for tools/ab_bench.sh and friends.
usage: build_variant.py name [flags ...]   (no flags: a copy of the current build)"""
import importlib
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
b = importlib.import_module("riv-slam_amd.build")
name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(b.HERE, f"_{name}.bin")
cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *b.FLAGS, f'-DAPD_SOURCE_STAMP="apd-source-stamp:{b.source_stamp()}"', *extra, "-o", out,
       *[os.path.join(b.CSRC, s) for s in b.SOURCES]]
subprocess.check_call(cmd)
print(out)
