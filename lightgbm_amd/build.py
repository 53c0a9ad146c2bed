"""Builds the native core in-tree.

Two shared libraries are produced next to this file:

* ``_lga_host.so`` - C++17/OpenMP: config, binning, host engine (CPU learner and
  test oracle), boosting drivers, text-model IO, host predictor, C ABI.
* ``_lga_hip.so``  - everything above plus the HIP (gfx950) kernels, the device
  engine and the RCCL communicator.  Cross-compiles without a GPU.

``python -m lightgbm_amd.build`` builds both; objects are cached under
``build/`` and only stale ones are recompiled.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
BUILD = PKG.parent / "build"

HOST_SOURCES = [
    "config.cpp", "bin_mapper.cpp", "dataset.cpp", "tree.cpp", "grower.cpp", "host_engine.cpp",
    "metric.cpp", "ranking.cpp", "gbdt.cpp", "model_io.cpp", "c_api.cpp",
]
HOST_ONLY = ["engine_factory_host.cpp"]
HIP_SOURCES = [
    "hip/kernels.hip", "hip/tree_kernel.hip", "hip/hip_engine.hip", "hip/predict.hip", "hip/binning.hip", "hip/rccl_comm.cpp",
    "hip/engine_factory_hip.cpp", "hip/hip_api.hip", "hip/ops.hip", "hip/loopback_comm.hip", "hip/bin_find.hip", "hip/sort.hip", "hip/auc.hip", "hip/ranking.hip", "hip/locality.hip",
]

COMMON_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-parameter",
                "-Wno-unused-function"]


def _headers_mtime() -> float:
    m = 0.0
    for p in list(CSRC.glob("*.hpp")) + list((CSRC / "hip").glob("*.hpp")) + list((CSRC / "hip").glob("*.h")):
        m = max(m, p.stat().st_mtime)
    return m


def _stale(src: Path, obj: Path, hdr_m: float) -> bool:
    if not obj.exists():
        return True
    om = obj.stat().st_mtime
    return om < src.stat().st_mtime or om < hdr_m


def _run(cmd, verbose):
    if verbose:
        print(" ".join(str(c) for c in cmd), flush=True)
    r = subprocess.run([str(c) for c in cmd], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(str(c) for c in cmd), r.stdout, r.stderr))
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)


def _compile_all(jobs, verbose, workers):
    if not jobs:
        return
    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(lambda c: _run(c, verbose), jobs))


def build_host(verbose: bool = False, force: bool = False) -> Path:
    cxx = os.environ.get("CXX", "g++")
    out = PKG / "_lga_host.so"
    objdir = BUILD / "host"
    objdir.mkdir(parents=True, exist_ok=True)
    hdr_m = _headers_mtime()
    jobs, objs = [], []
    for s in HOST_SOURCES + HOST_ONLY:
        src = CSRC / s
        obj = objdir / (s.replace("/", "_") + ".o")
        objs.append(obj)
        if force or _stale(src, obj, hdr_m):
            jobs.append([cxx, *COMMON_FLAGS, "-fopenmp", "-I", CSRC, "-c", src, "-o", obj])
    _compile_all(jobs, verbose, workers=min(8, os.cpu_count() or 1))
    if jobs or not out.exists():
        _run([cxx, "-shared", "-fopenmp", "-o", out, *objs], verbose)
    return out


def hipcc_path() -> str | None:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and Path(c).exists():
            return c
    return None


def build_hip(verbose: bool = False, force: bool = False, debug: bool = False) -> Path:
    """debug=True builds ``_lga_hip_debug.so``: the same core with -DLGA_DEBUG, whose kernels check every
    scattered index before use and report a violation instead of faulting (``LGA_DEBUG_LIB=1`` loads it)."""
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found; cannot build the HIP core")
    out = PKG / ("_lga_hip_debug.so" if debug else "_lga_hip.so")
    objdir = BUILD / ("hip_debug" if debug else "hip")
    objdir.mkdir(parents=True, exist_ok=True)
    hdr_m = _headers_mtime()
    flags = [*COMMON_FLAGS, "--offload-arch=gfx950", "-fopenmp", "-DLGA_WITH_HIP=1", "-I", CSRC, "-I", CSRC / "hip",
             "-I", "/opt/rocm/include"]
    if debug:
        flags.append("-DLGA_DEBUG=1")
    jobs, objs = [], []
    for s in HOST_SOURCES + HIP_SOURCES:
        src = CSRC / s
        obj = objdir / (s.replace("/", "_") + ".o")
        objs.append(obj)
        if force or _stale(src, obj, hdr_m):
            extra = ["-x", "hip"] if s.endswith(".hip") else []
            jobs.append([hipcc, *flags, *extra, "-c", src, "-o", obj])
    _compile_all(jobs, verbose, workers=min(8, os.cpu_count() or 1))
    if jobs or not out.exists():
        _run([hipcc, "-shared", "-fopenmp", "--offload-arch=gfx950", "-o", out, *objs, "-ldl"], verbose)
    return out


def build_all(verbose: bool = False, force: bool = False):
    return build_host(verbose, force), build_hip(verbose, force), build_hip(verbose, force, debug=True)


if __name__ == "__main__":
    force = "--force" in sys.argv
    which = [a for a in sys.argv[1:] if not a.startswith("-")]
    if not which or "host" in which:
        print(build_host(True, force))
    if not which or "hip" in which:
        print(build_hip(True, force))
    if not which or "debug" in which:
        print(build_hip(True, force, debug=True))
