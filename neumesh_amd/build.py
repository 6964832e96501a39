"""Build libneumesh_hip.so for gfx950 with hipcc (in-tree, next to the sources).

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(CSRC, "libneumesh_hip.so")
# the same sources with -DNM_TESTING: the product's exports + test hooks (host octree build, octree export, scalar-ALU self-check
# of the MFMA tile code, phase stamps).  Loaded by a few tests and by the measurement tools only -- never by the package.
TESTING_LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(CSRC)), "tests", "_build", "libneumesh_hip_testing.so")
SOURCES = ["nm_api.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))) + [os.path.join("..", "..", "include", "neumesh_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-unused-value"]


def _stale(path: str) -> bool:
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def _compile(path: str, extra, verbose: bool) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build " + os.path.basename(path))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    cmd = [hipcc, *FLAGS, *extra, *[os.path.join(CSRC, s) for s in SOURCES], "-o", path + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(path + ".tmp", path)
    return path


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP library if it is missing or older than its sources. Returns its path."""
    if not force and not _stale(LIB_PATH):
        return LIB_PATH
    return _compile(LIB_PATH, [], verbose)


def build_testing(force: bool = False, verbose: bool = False) -> str:
    """The test / measurement build (-DNM_TESTING) under tests/_build/.  Returns its path."""
    if not force and not _stale(TESTING_LIB_PATH):
        return TESTING_LIB_PATH
    return _compile(TESTING_LIB_PATH, ["-DNM_TESTING"], verbose)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_testing(force=True, verbose=True))
