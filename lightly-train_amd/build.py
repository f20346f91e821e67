"""Builds liblt_amd.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
Objects are cached by source mtime; the .so is git-ignored but travels with gpurun snapshots.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(LIBDIR, "liblt_amd.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")


def sources() -> list[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _newest_header() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdr_t = _newest_header()
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            cmd = [hipcc, *FLAGS, "-x", "hip", "-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd: list[str]) -> None:
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


def build_timing() -> str:
    """Diagnostic variant for tools/gemm_timeline.py: gemm.hip with -DLT_GEMM_TIMING (per-workgroup timestamps of the four-phase GEMM
    kernel and the debug accessor lt_debug_gemm_timing), linked with the shipped objects as lib/liblt_amd_timing.so."""
    build()
    hipcc = _hipcc()
    obj = os.path.join(OBJDIR, "gemm_timing.o")
    out = os.path.join(os.path.dirname(LIB), "liblt_amd_timing.so")
    src = [s_ for s_ in sources() if s_.endswith("gemm.hip")][0]
    subprocess.run([hipcc, *FLAGS, "-DLT_GEMM_TIMING", "-x", "hip", "-c", src, "-o", obj], check=True)
    others = [os.path.join(OBJDIR, os.path.basename(s_) + ".o") for s_ in sources() if not s_.endswith("gemm.hip")]
    subprocess.run([hipcc, FLAGS[0], "-shared", "-fPIC", *others, obj, "-o", out], check=True)
    return out


def build_variant(defines: list, name: str) -> str:
    """gemm.hip rebuilt with extra -D flags, linked with the shipped objects as lib/liblt_amd_<name>.so (tools/ab_lib.py alternates it with the
    shipped library step by step in one process)."""
    build()
    hipcc = _hipcc()
    obj = os.path.join(OBJDIR, f"gemm_{name}.o")
    out = os.path.join(os.path.dirname(LIB), f"liblt_amd_{name}.so")
    src = [s_ for s_ in sources() if s_.endswith("gemm.hip")][0]
    subprocess.run([hipcc, *FLAGS, *[f"-D{d}" for d in defines], "-x", "hip", "-c", src, "-o", obj], check=True)
    others = [os.path.join(OBJDIR, os.path.basename(s_) + ".o") for s_ in sources() if not s_.endswith("gemm.hip")]
    subprocess.run([hipcc, FLAGS[0], "-shared", "-fPIC", *others, obj, "-o", out], check=True)
    return out


if __name__ == "__main__":
    import sys
    print(build_timing() if "--timing" in sys.argv else build(verbose=True))
