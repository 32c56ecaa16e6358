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
# (the stamp compiled in covers the extra flags: a variant never carries the product's stamp, and the library lists its experiment defines
# itself -- apdgicp_build_flags(); the loader refuses such a library unless APDGICP_ALLOW_VARIANT_LIB=1)
subprocess.check_call(b.compile_command(out, extra))
print(out)
