"""Builds libapdgicp_hip.so (HIP kernels + C ABI) in-tree for gfx950.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libapdgicp_hip.so")
SOURCES = ["apdgicp_hip.hip"]
DEPS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))) + [os.path.join("..", "..", "include", "apdgicp_hip.h")]  # every source the one translation unit includes
FLAGS = [*os.environ.get("APD_EXTRA_FLAGS", "").split(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wno-unused-result"]


def source_stamp() -> str:
    """Fingerprint of the kernel sources (csrc/*.hip, *.hpp): profiles/pmc_nn_latest.json carries the stamp of the sources its
    counters were collected from (tools/pmc_nn_json.py), and bench.py reports PMC-derived numbers only when it equals this one."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode() + b"\0")
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS if os.path.exists(os.path.join(CSRC, d)))


def build(force: bool = False, verbose: bool = False, extra: list[str] | None = None) -> str:
    """One process compiles at a time (several ranks of one node may call this together): an exclusive lock around the
    freshness check, output into a temporary file that is renamed over the library only when complete."""
    import fcntl
    if not force and not needs_build():
        return LIB
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or needs_build():  # (another process may have built it while this one waited)
                hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
                tmp = f"{LIB}.{os.getpid()}.tmp"
                cmd = [hipcc, *FLAGS, *(extra or []), "-o", tmp, *[os.path.join(CSRC, s) for s in SOURCES]]
                if verbose:
                    print(" ".join(cmd), file=sys.stderr)
                try:
                    subprocess.check_call(cmd)
                    os.replace(tmp, LIB)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
