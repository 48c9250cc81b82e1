"""Builds the two in-tree shared libraries of the package (no JIT cache: the .so files travel with the tree).

  lib/libykpred.so   HIP engine, gfx950 only        (csrc/engine/engine.hip  → include/ykpred.h)
  lib/libykhost.so   host-side mirror, plain C++17  (csrc/host/host.cpp      → include/ykhost.h), links libykpred.so

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container as well as on the GPU box.
"""
import contextlib
import fcntl
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "lib")
ENGINE_SRC = os.path.join(PKG, "csrc", "engine", "engine.hip")
ENGINE_DEPS = [ENGINE_SRC, os.path.join(PKG, "csrc", "engine", "kernels.hip.h"), os.path.join(ROOT, "include", "ykpred.h")]
HOST_SRC = os.path.join(PKG, "csrc", "host", "host.cpp")
HOST_DEPS = [HOST_SRC] + [os.path.join(PKG, "csrc", "host", f) for f in ("encoder.h", "objects.h", "minijson.h", "quantity.h", "jsonscan.h")] + [
    os.path.join(ROOT, "include", "ykhost.h"), os.path.join(ROOT, "include", "ykpred.h")]
LIBYKPRED = os.path.join(LIB, "libykpred.so")
LIBYKHOST = os.path.join(LIB, "libykhost.so")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# -ffp-contract=off: the float64 bin-pack score must match the oracle's operation order bit for bit.
HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-Wextra",
            "-Wno-unused-parameter"]
CXXFLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-Wno-unused-parameter"]


@contextlib.contextmanager
def _build_lock():
    """One builder at a time: bench.py is launched as N ranks that all import the package at once."""
    os.makedirs(LIB, exist_ok=True)
    with open(os.path.join(LIB, ".build.lock"), "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def _compile(cmd, target, verbose):
    """Compile to a temporary name and rename: a concurrently starting process never maps a half-written library."""
    tmp = target + f".tmp{os.getpid()}"
    cmd = [tmp if c == target else c for c in cmd]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, target)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_engine(force=False, verbose=False):
    with _build_lock():
        if force or _stale(LIBYKPRED, ENGINE_DEPS):
            # (YKPRED_EXTRA_HIPFLAGS: compile-time knobs of kernel experiments, e.g. -DYK_SWEEP_BATCH=4 — scripts/r06_*.sh)
            _compile([HIPCC] + HIPFLAGS + os.environ.get("YKPRED_EXTRA_HIPFLAGS", "").split() + [ENGINE_SRC, "-o", LIBYKPRED], LIBYKPRED, verbose)
    return LIBYKPRED


def build_host(force=False, verbose=False):
    build_engine(force=False, verbose=verbose)
    with _build_lock():
        if force or _stale(LIBYKHOST, HOST_DEPS + [LIBYKPRED]):
            _compile(["g++"] + CXXFLAGS + [HOST_SRC, "-o", LIBYKHOST, "-L" + LIB, "-lykpred", "-Wl,-rpath,$ORIGIN"], LIBYKHOST, verbose)
    return LIBYKHOST


def build_all(force=False, verbose=False):
    build_engine(force, verbose)
    build_host(force, verbose)
    return LIBYKPRED, LIBYKHOST


if __name__ == "__main__":
    import sys
    build_all(force="--force" in sys.argv, verbose=True)
