"""Build libwavjepa_hip.so (gfx950) in-tree with hipcc.  No torch dependency: plain HIP + a C ABI."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libwavjepa_hip.so")
LAB_LIB = os.path.join(LIBDIR, "libwavjepa_hip_lab.so")      # the same sources with -DWJ_LAB (include/wavjepa_hip_lab.h): diagnostics + A/B switches
SOURCES = ["gemm.hip", "gemm_persist.hip", "gemm_pde.hip", "gemm_panel.hip", "norm.hip", "attention.hip", "conv0.hip", "misc.hip", "fp8.hip", "scene.hip", "denoise.hip", "rccl_bucket.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP toolchain (ROCm) is required to build libwavjepa_hip.so")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, lab: bool = False) -> str:
    """The release library (compute + query entries, production-safe switches only), or with lab=True the laboratory variant
    (libwavjepa_hip_lab.so: -DWJ_LAB, the extra entries of include/wavjepa_hip_lab.h and every A/B / diagnostic environment switch)."""
    hipcc = _hipcc()
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj_lab" if lab else "obj")
    os.makedirs(objdir, exist_ok=True)
    inc = os.path.join(os.path.dirname(HERE), "include")
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_internal.h"), os.path.join(inc, "wavjepa_hip.h"), os.path.join(inc, "wavjepa_hip_lab.h")]
    flags = FLAGS + (["-DWJ_LAB"] if lab else [])
    lib = LAB_LIB if lab else LIB
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + flags + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr}")
        return s

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for s in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[wavjepa_amd.build] compiled {os.path.basename(s)}", file=sys.stderr)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[wavjepa_amd.build] linked {lib}", file=sys.stderr)
    if not lab:
        build_io(force=force, verbose=verbose)
    return lib


def build_all(force: bool = False, verbose: bool = True) -> None:
    """Release + laboratory libraries (what __graft_entry__.build() and the test suite need)."""
    build(force=force, verbose=verbose)
    build(force=force, verbose=verbose, lab=True)


IO_LIB = os.path.join(LIBDIR, "libwavjepa_io.so")
IO_SOURCES = ["flac_decode.cpp"]


def build_io(force: bool = False, verbose: bool = True) -> str:
    """Host-side library of the data path (FLAC decoder): plain C++ with a C ABI, built with g++ (no GPU code)."""
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in IO_SOURCES]
    if force or _stale(IO_LIB, srcs):
        cxx = shutil.which("g++") or shutil.which("c++")
        if cxx is None:
            raise RuntimeError("g++ not found: needed for libwavjepa_io.so")
        r = subprocess.run([cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", IO_LIB] + srcs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"g++ failed for libwavjepa_io.so:\n{r.stderr}")
        if verbose:
            print(f"[wavjepa_amd.build] built {IO_LIB}", file=sys.stderr)
    return IO_LIB


if __name__ == "__main__":
    if "--release-only" in sys.argv:
        build(force="--force" in sys.argv)
    else:
        build_all(force="--force" in sys.argv)
