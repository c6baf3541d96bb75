"""Build libspfsplat_hip.so (gfx950 only) in-tree with hipcc.

    python -m spfsplatv2_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the tree.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OUT_DIR = PKG / os.environ.get("SPF_LIB_DIR", "_C")      # SPF_LIB_DIR=_C_xyz: a variant build next to the regular one
LIB = OUT_DIR / "libspfsplat_hip.so"
SOURCES = ["api.hip", "adapter.hip", "camera.hip", "project.hip", "binning.hip", "render.hip", "rope2d.hip", "loss.hip"]
HEADERS = [CSRC / "spf_common.h", PKG.parent / "include" / "spfsplat_hip.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("SPF_HIPCC_EXTRA", "").split()      # profiling builds, e.g. -DSPF_PHASE_CLOCKS


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _digest() -> str:
    h = hashlib.sha256()
    for p in [CSRC / s for s in SOURCES] + HEADERS:
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> Path:
    OUT_DIR.mkdir(exist_ok=True)
    stamp = OUT_DIR / "build.sha256"
    dig = _digest()
    if not force and LIB.exists() and stamp.exists() and stamp.read_text().strip() == dig:
        return LIB
    hipcc = _hipcc()

    def compile_one(src: str) -> Path:
        obj = OUT_DIR / (src + ".o")
        cmd = [hipcc, *FLAGS, "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    stamp.write_text(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
