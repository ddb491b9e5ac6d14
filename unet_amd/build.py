"""Builds ``unet_amd/lib/libunet_mi355.so`` from the HIP sources with hipcc for gfx950.

The library is built in-tree (cross-compiled on a machine without a GPU if need be); nothing is
compiled at import time.  ``python -m unet_amd.build`` or ``__graft_entry__.build()`` runs it.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIBDIR = HERE / "lib"
OBJDIR = HERE / "lib" / "obj"
LIBNAME = "libunet_mi355.so"
SOURCES = ["igemm.hip", "igemm_dma.hip", "conv2d.hip", "wgrad2.hip", "wgrad3.hip", "norm.hip", "small.hip", "loss.hip", "pipeline.hip", "ops.hip", "engine.hip", "comm.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-munsafe-fp-atomics"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm is required to build the HIP kernels)")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    hipcc = _hipcc()
    LIBDIR.mkdir(exist_ok=True)
    OBJDIR.mkdir(exist_ok=True)
    headers = sorted(CSRC.glob("*.h")) + [HERE.parent / "include" / "unet_mi355.h"]
    jobs = []
    for src in SOURCES:
        obj = OBJDIR / (src + ".o")
        if force or _stale(obj, [CSRC / src] + headers):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-c", str(CSRC / src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for src, rc, out in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[unet_amd.build] hipcc {src}: {'ok' if rc == 0 else 'FAILED'}", flush=True)
                if rc != 0:
                    raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    lib = LIBDIR / LIBNAME
    objs = [OBJDIR / (s + ".o") for s in SOURCES]
    if force or jobs or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [str(o) for o in objs] + ["-o", str(lib)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        if verbose:
            print(f"[unet_amd.build] linked {lib}", flush=True)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
