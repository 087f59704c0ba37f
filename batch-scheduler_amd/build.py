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
# translation units -> the headers each one depends on (bs_launch.hpp is the interface between them; DESIGN.md section 4 "Build")
_COMMON = ["bs_common.hpp", "bs_kernels.hpp", "bs_launch.hpp", os.path.join("..", "..", "include", "bsched.h")]
UNITS = {
    "bsched.hip": _COMMON + ["bs_fast.hpp", "bs_filter_t.hpp", "bs_epoch.hpp", "bs_queue.hpp", "bs_fdeny.hpp", "bs_seq.hpp", "bs_sort.hpp", "bs_fit.hpp"],
    "tu_fast.hip": _COMMON + ["bs_fast.hpp", "bs_filter_t.hpp"],
    "tu_seq.hip": _COMMON + ["bs_seq.hpp"],
}
SOURCES = list(UNITS)
HEADERS = sorted({h for hs in UNITS.values() for h in hs})
OBJ_DIR = os.path.join(HERE, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the gfx950 library cannot be built")
    return exe


HOST_LIB_PATH = os.path.join(LIB_DIR, "libbsched_host.so")
HOST_SRC = os.path.join(HERE, "host", "bs_host.cpp")
HOST_SRCS = [HOST_SRC, os.path.join(HERE, "host", "bs_drain.cpp"), os.path.join(HERE, "host", "bs_phase.cpp")]


def build_host(force: bool = False, verbose: bool = False) -> str:
    """The C++ host-side mirror of the reference's ScheduleOperation; links against libbsched.so."""
    build()
    if PREBUILT_ONLY:
        return HOST_LIB_PATH
    deps = [*HOST_SRCS, os.path.join(HERE, "..", "include", "bsched.h"), os.path.join(HERE, "..", "include", "bsched_host.h"), LIB_PATH]
    if not force and os.path.exists(HOST_LIB_PATH) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_LIB_PATH) for d in deps):
        return HOST_LIB_PATH
    cxx = shutil.which("g++") or "g++"
    tmp = f"{HOST_LIB_PATH}.{os.getpid()}.tmp"          # linked beside the target, then renamed: another process (pytest -n) never loads half a file
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", tmp, *HOST_SRCS, "-L" + HERE, "-lbsched",
           "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("g++ failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, HOST_LIB_PATH)
    return HOST_LIB_PATH


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def _obj(src: str) -> str:
    return os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")


def _unit_stale(src: str) -> bool:
    o = _obj(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in [src] + UNITS[src])


def build(force: bool = False, verbose: bool = False, extra_flags: list[str] | None = None, unity: bool = False) -> str:
    """Compiles the translation units that are out of date (in parallel) and links libbsched.so.  extra_flags (probe / experiment builds)
    always rebuild everything; unity=True compiles bsched.hip alone with -DBS_UNITY (it then includes the other units: the probe builds
    need their __device__ probe arrays in ONE translation unit)."""
    if PREBUILT_ONLY:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"BS_LIB_DIR is set but {LIB_PATH} does not exist")
        return LIB_PATH
    if not force and not extra_flags and not is_stale():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    if unity:
        cmd = [hipcc(), *FLAGS, "-shared", "-DBS_UNITY", *(extra_flags or []), "-o", LIB_PATH, os.path.join(CSRC, "bsched.hip"), "-ldl"]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
        for src in SOURCES:                                   # the objects no longer match the library
            if os.path.exists(_obj(src)):
                os.remove(_obj(src))
        return LIB_PATH
    todo = [src for src in SOURCES if force or extra_flags or _unit_stale(src)]
    procs = []
    for src in todo:
        cmd = [hipcc(), *FLAGS, *(extra_flags or []), "-c", "-o", _obj(src), os.path.join(CSRC, src)]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            failed.append(f"{src}:\n{out}")
        elif verbose and out:
            print(out)
    if failed:
        for src, _ in procs:                                  # never leave a half-built set that looks fresh
            if os.path.exists(_obj(src)) and any(f.startswith(src) for f in failed):
                os.remove(_obj(src))
        raise RuntimeError("hipcc failed:\n" + "\n".join(failed))
    tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *[_obj(src) for src in SOURCES], "-ldl"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB_PATH)
    if extra_flags:
        for src in SOURCES:                                   # objects of an experiment build must not be taken for the shipped ones
            os.remove(_obj(src))
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
