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


BINDING_SRC = CSRC / "torch_binding.cpp"
BINDING = OUT_DIR / "_spf_torch.so"


def build_binding(force: bool = False, verbose: bool = True) -> Path:
    """The compiled host binding for PyTorch callers (csrc/torch_binding.cpp -> _spf_torch.so next to the library):
    plain g++ against the installed torch's headers -- host code only, no device code, no hipify.  Optional at run time
    (rasterizer.py falls back to its ctypes path), so a failure here is reported by the caller, not hidden."""
    import sysconfig

    import torch
    tdir = Path(torch.__file__).resolve().parent
    stamp = OUT_DIR / "binding.sha256"
    h = hashlib.sha256()
    for p in (BINDING_SRC, HEADERS[1]):
        h.update(p.read_bytes())
    h.update(torch.__version__.encode())
    dig = h.hexdigest()
    if not force and BINDING.exists() and stamp.exists() and stamp.read_text().strip() == dig:
        return BINDING
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", str(BINDING_SRC), "-o", str(BINDING),
           f"-I{HEADERS[1].parent}", f"-I{tdir / 'include'}", f"-I{tdir / 'include' / 'torch' / 'csrc' / 'api' / 'include'}",
           f"-I{sysconfig.get_paths()['include']}", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=_spf_torch", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch.compiled_with_cxx11_abi())}", "-w",
           f"-L{tdir / 'lib'}", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_python",
           f"-L{OUT_DIR}", "-lspfsplat_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,$ORIGIN",
           f"-Wl,-rpath,{tdir / 'lib'}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    stamp.write_text(dig)
    return BINDING


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_binding(force="--force" in sys.argv))
