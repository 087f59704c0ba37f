"""Builds the in-tree HIP shared library (gfx950 only) with hipcc.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.  There is no JIT and no
fallback: if hipcc is missing or the build fails this raises.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# BS_LIB_DIR: load libbsched.so / libbsched_host.so from another directory and never rebuild them (the sanitizer builds of
# tools/build_sanitized.sh live in their own directory under the same file names, so that $ORIGIN resolves the pair)
LIB_DIR = os.environ.get("BS_LIB_DIR") or HERE
PREBUILT_ONLY = bool(os.environ.get("BS_LIB_DIR"))
LIB_PATH = os.path.join(LIB_DIR, "libbsched.so")
SOURCES = ["bsched.hip"]
HEADERS = ["bs_common.hpp", "bs_kernels.hpp", "bs_fast.hpp", "bs_filter_t.hpp", "bs_epoch.hpp", "bs_queue.hpp", "bs_fdeny.hpp", "bs_seq.hpp", "bs_sort.hpp", "bs_fit.hpp", os.path.join("..", "..", "include", "bsched.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the gfx950 library cannot be built")
    return exe


HOST_LIB_PATH = os.path.join(LIB_DIR, "libbsched_host.so")
HOST_SRC = os.path.join(HERE, "host", "bs_host.cpp")
HOST_SRCS = [HOST_SRC, os.path.join(HERE, "host", "bs_drain.cpp")]


def build_host(force: bool = False, verbose: bool = False) -> str:
    """The C++ host-side mirror of the reference's ScheduleOperation; links against libbsched.so."""
    build()
    if PREBUILT_ONLY:
        return HOST_LIB_PATH
    deps = [*HOST_SRCS, os.path.join(HERE, "..", "include", "bsched.h"), LIB_PATH]
    if not force and os.path.exists(HOST_LIB_PATH) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_LIB_PATH) for d in deps):
        return HOST_LIB_PATH
    cxx = shutil.which("g++") or "g++"
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", HOST_LIB_PATH, *HOST_SRCS, "-L" + HERE, "-lbsched",
           "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("g++ failed:\n" + res.stdout + res.stderr)
    return HOST_LIB_PATH


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags: list[str] | None = None) -> str:
    if PREBUILT_ONLY:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"BS_LIB_DIR is set but {LIB_PATH} does not exist")
        return LIB_PATH
    if not force and not is_stale():
        return LIB_PATH
    cmd = [hipcc(), *FLAGS, *(extra_flags or []), "-o", LIB_PATH, *[os.path.join(CSRC, s) for s in SOURCES], "-ldl"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    if verbose and res.stderr:
        print(res.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
