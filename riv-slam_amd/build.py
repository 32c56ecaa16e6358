"""Builds libapdgicp_hip.so (HIP kernels + C ABI) in-tree for gfx950.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libapdgicp_hip.so")
SOURCES = ["apdgicp_hip.hip"]
INCLUDE = os.path.join(HERE, "..", "include")
HEADERS = ["apdgicp_hip.h", "apd_atan2f.h"]  # include/: what the one translation unit includes from there
FLAGS = [*os.environ.get("APD_EXTRA_FLAGS", "").split(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wno-unused-result"]


def source_stamp(extra: list[str] | None = None) -> str:
    """Fingerprint of everything the library is compiled from (csrc/*.hip, *.hpp, the two headers under include/, the compiler
    flags -- `extra` ones included: a variant built with -DAPD_ABL_* never carries the product's stamp).  It is compiled INTO the library (apdgicp_source_stamp()): the loader, the test suite and bench.py compare the
    loaded library's stamp with this one, and profiles/pmc_nn_latest.json carries the stamp of the library its counters were
    collected from (tools/pmc_nn_json.py) -- bench.py reports PMC-derived numbers only when all three agree."""
    import hashlib
    h = hashlib.sha256()
    for d, names in ((CSRC, sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp")))), (INCLUDE, HEADERS)):
        for f in names:
            h.update(f.encode() + b"\0")
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    if extra:
        h.update(b"\0extra\0" + " ".join(extra).encode())
    return h.hexdigest()[:16]


def compile_command(out: str, extra: list[str] | None = None) -> list[str]:
    """The one hipcc command line (build() and tools/build_variant.py): flags, the stamp of sources + flags + extra flags, and the flags
    themselves as a string the library returns from apdgicp_build_flags()."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = list(extra or [])
    shown = " ".join(FLAGS + extra).replace('"', "'")
    return [hipcc, *FLAGS, f'-DAPD_SOURCE_STAMP="apd-source-stamp:{source_stamp(extra)}"', f'-DAPD_BUILD_FLAGS="{shown}"', *extra, "-o", out,
            *[os.path.join(CSRC, s) for s in SOURCES]]


def library_stamp(path: str = LIB) -> str | None:
    """The stamp compiled into a built library, read from the file itself (no dlopen: the bytes behind the marker)."""
    if not os.path.exists(path):
        return None
    import re
    with open(path, "rb") as fh:
        m = re.search(rb"apd-source-stamp:([0-9a-f]{16})", fh.read())
    return m.group(1).decode() if m else None


def needs_build() -> bool:
    """Identity, not age: the library is current when the stamp inside it equals the stamp of the sources on disk."""
    return library_stamp() != source_stamp()


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
                tmp = f"{LIB}.{os.getpid()}.tmp"
                cmd = compile_command(tmp, extra)
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
