"""Generate RoPE-2D golden vectors from the REFERENCE's own implementations (build container only).

  * the pure-PyTorch fallback ``RoPE2D`` (/root/reference/src/model/encoder/backbone/croco/pos_embed.py:112-159),
    imported by file path (its ``from .curope import cuRoPE2D`` fails -> the reference selects the fallback itself);
  * the reference's C++ CPU loop ``rope_2d`` (croco/curope/curope.cpp:11-65) from oracle/_ref/curope_ref.so when
    it has been built (`make -C oracle ref`), loaded with RTLD_LAZY because its CUDA half does not exist here.

Writes tests/golden/rope_goldens.pt.      python tests/golden/make_rope_goldens.py
"""
import importlib.util
import os
import sys
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parents[1]
REF = Path("/root/reference/src/model/encoder/backbone/croco")


def load_fallback():
    spec = importlib.util.spec_from_file_location("ref_pos_embed", REF / "pos_embed.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)          # prints the reference's own "cannot find cuda-compiled version" warning
    return m.RoPE2D


def load_cpp():
    so = REPO / "oracle" / "_ref" / "curope_ref.so"
    if not so.exists():
        return None
    old = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_LOCAL)
    try:
        spec = importlib.util.spec_from_file_location("curope_ref", so)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        sys.setdlopenflags(old)
    return m


def positions(gen, B, hh, ww, extra):
    """(y, x) grid positions as PositionGetter produces them (croco/blocks.py:207-219) plus `extra` tokens at
    (y_last+1+i, 0) like the intrinsics / pose tokens (backbone_masked_croco.py:163-172)."""
    y, x = torch.meshgrid(torch.arange(hh), torch.arange(ww), indexing="ij")
    pos = torch.stack([y.reshape(-1), x.reshape(-1)], dim=-1)
    for i in range(extra):
        pos = torch.cat([pos, torch.tensor([[hh + i, 0]])])
    return pos[None].expand(B, -1, -1).clone().long()


def main():
    RoPE2D = load_fallback()
    cpp = load_cpp()
    gen = torch.Generator().manual_seed(3)
    cases = {}
    for name, (B, hh, ww, extra, H, D) in {
        "dec_38x12x64": (2, 6, 6, 2, 12, 64),       # decoder layout: patches + intrinsics + pose token
        "enc_64x16x64": (1, 8, 8, 0, 16, 64),
        "far_40x4x64": (1, 2, 20, 0, 4, 64),        # x positions up to 19 (256x256 input reaches 17)
        "odd_15x3x24": (3, 5, 3, 0, 3, 24),         # Q = 6: not a multiple of 4 -> scalar kernel path
        "d128_20x2x128": (1, 4, 5, 0, 2, 128),
    }.items():
        pos = positions(gen, B, hh, ww, extra)
        N = pos.shape[1]
        tok = torch.randn(B, H, N, D, generator=gen)           # module layout [B,H,N,D]
        rope = RoPE2D(freq=100.0, F0=1.0)
        out_py = rope(tok.clone(), pos)
        entry = {"tokens_BHND": tok, "positions": pos, "base": 100.0, "F0": 1.0, "out_fallback_BHND": out_py}
        if cpp is not None:
            t = tok.clone().transpose(1, 2).contiguous()       # [B,N,H,D] contiguous for the CPU accessor
            cpp.rope_2d(t, pos, 100.0, 1.0)
            entry["out_cpp_BNHD"] = t.clone()
            cpp.rope_2d(t, pos, 100.0, -1.0)
            entry["roundtrip_cpp_maxerr"] = float((t - tok.transpose(1, 2)).abs().max())
            print(name, "max|fallback - cpp| =", (out_py.transpose(1, 2) - entry["out_cpp_BNHD"]).abs().max().item())
        cases[name] = entry
    # VGGT's RotaryPositionEmbedding2D (vggt/layers/rope.py:62-188): out of place, head-major [B,H,N,D]
    spec = importlib.util.spec_from_file_location("ref_vggt_rope", REF.parent / "vggt" / "layers" / "rope.py")
    vm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(vm)
    vggt = {}
    for name, (B, H, hh, ww, D) in {"vggt_2x4_37x64": (2, 4, 6, 6, 64), "vggt_1x16_50x128": (1, 16, 5, 10, 128)}.items():
        pos = vm.PositionGetter()(B, hh, ww, torch.device("cpu")) + 1          # VGGT offsets patch positions by 1 ...
        pos = torch.cat([torch.zeros(B, 1, 2, dtype=pos.dtype), pos], dim=1)     # ... and gives special tokens (0, 0)
        tok = torch.randn(B, H, pos.shape[1], D, generator=gen)
        out = vm.RotaryPositionEmbedding2D(frequency=100.0)(tok, pos)
        vggt[name] = {"tokens_BHND": tok, "positions": pos, "frequency": 100.0, "out_BHND": out}
    torch.save({"cases": cases, "has_cpp": cpp is not None, "vggt": vggt}, HERE / "rope_goldens.pt")
    print("wrote", HERE / "rope_goldens.pt", "cpp:", cpp is not None)


if __name__ == "__main__":
    main()
