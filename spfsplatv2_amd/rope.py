"""2-D rotary position embedding on the HIP library: host mirror of the reference's ``curope`` package.

=====================  ===================================================================================
here                   reference
=====================  ===================================================================================
``rope_2d``            ``curope.rope_2d`` (pybind), croco/curope/curope.cpp:49-69 + curope/kernels.cu:84-108
``cuRoPE2D_func``      croco/curope/curope2d.py:12-29
``cuRoPE2D``           croco/curope/curope2d.py:32-40  (selected by croco/pos_embed.py:106-110)
=====================  ===================================================================================
(paths under /root/reference/src/model/encoder/backbone/)

Same contract: in place, tokens[B,N,H,D] may be a strided view as long as ``stride(3) == 1`` and
``stride(2) == D``; positions[B,N,2] int64 contiguous; error messages follow the reference's checks.
There is no CPU path here (the reference's CPU loop lives on as the test oracle only).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

_DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _check(tokens: torch.Tensor, positions: torch.Tensor) -> None:
    """The reference's argument checks (curope.cpp:54-59, kernels.cu:91-94), same messages."""
    if tokens.dim() != 4:
        raise RuntimeError("tokens must have 4 dimensions")
    if positions.dim() != 3:
        raise RuntimeError("positions must have 3 dimensions")
    if tokens.size(0) != positions.size(0):
        raise RuntimeError("batch size differs between tokens & positions")
    if tokens.size(1) != positions.size(1):
        raise RuntimeError("seq_length differs between tokens & positions")
    if positions.size(2) != 2:
        raise RuntimeError("positions.shape[2] must be equal to 2")
    if tokens.is_cuda != positions.is_cuda:
        raise RuntimeError("tokens and positions are not on the same device")
    if not tokens.is_cuda:
        raise RuntimeError("rope_2d: tokens are on the CPU; this build only runs on a HIP device "
                           "(no CPU fallback)")
    D = tokens.shape[3]
    if not (tokens.stride(3) == 1 and tokens.stride(2) == D):
        raise RuntimeError("tokens are not contiguous")
    if not positions.is_contiguous():
        raise RuntimeError("positions are not contiguous")
    if positions.dtype != torch.int64:
        raise RuntimeError("positions must be int64")
    if D % 4 != 0:
        raise RuntimeError("token dim must be multiple of 4")
    if tokens.dtype not in _DTYPES:
        raise RuntimeError(f"rope_2d: unsupported dtype {tokens.dtype}")


def rope_2d(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> None:
    """In-place RoPE-2D on ``tokens`` (B,N,H,D); ``fwd`` = +F0 forward, -F0 backward."""
    _check(tokens, positions)
    B, N, H, D = tokens.shape
    _launch(tokens, positions, B, N, H, D, tokens.stride(0), tokens.stride(1), D, 1, base, fwd)


def rope_2d_pair(q: torch.Tensor, k: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> None:
    """``rope_2d`` on two tensors of the same shape, strides and dtype in ONE launch -- q and k of an attention layer,
    typically two views of one qkv buffer (croco/blocks.py:97-104 rotates them with two calls): the angles are
    evaluated once.  Falls back to two calls when the layouts differ."""
    if q.shape != k.shape or q.stride() != k.stride() or q.dtype != k.dtype or q.device != k.device:
        rope_2d(q, positions, base, fwd)
        rope_2d(k, positions, base, fwd)
        return
    _check(q, positions)
    B, N, H, D = q.shape
    lib = _lib.load()
    with torch.cuda.device(q.device):
        stream = C.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)
        _lib.check(lib.spf_rope2d_pair(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()),
                                       C.c_void_p(positions.data_ptr()), B, N, H, D, q.stride(0), q.stride(1), D, 1,
                                       _DTYPES[q.dtype], float(base), float(fwd), stream), "spf_rope2d_pair")


def _launch(tokens, positions, B, N, H, D, sb, sn, sh, pos_div, base, fwd) -> None:
    lib = _lib.load()
    with torch.cuda.device(tokens.device):      # HIP launches go to the current device
        stream = C.c_void_p(torch.cuda.current_stream(tokens.device).cuda_stream)
        _lib.check(lib.spf_rope2d(C.c_void_p(tokens.data_ptr()), C.c_void_p(positions.data_ptr()), B, N, H, D, sb, sn,
                                  sh, pos_div, _DTYPES[tokens.dtype], float(base), float(fwd), stream), "spf_rope2d")


def rope_2d_head_major(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> None:
    """In-place RoPE-2D on a CONTIGUOUS head-major tensor [B,H,N,D] (VGGT's layout): every (batch, head) slab is
    one batch item of one head that shares batch b's positions."""
    if tokens.dim() != 4 or not tokens.is_contiguous():
        raise RuntimeError("tokens must be a contiguous [B,H,N,D] tensor")
    if positions.dim() != 3 or positions.shape[-1] != 2 or positions.shape[0] != tokens.shape[0] or \
            positions.shape[1] != tokens.shape[2]:
        raise RuntimeError("Positions must have shape (batch_size, n_tokens, 2)")
    if not tokens.is_cuda or not positions.is_cuda:
        raise RuntimeError("rope_2d: tensors are on the CPU; this build only runs on a HIP device (no CPU fallback)")
    B, H, N, D = tokens.shape
    if D % 4 != 0:
        raise RuntimeError("token dim must be multiple of 4")
    if tokens.dtype not in _DTYPES:
        raise RuntimeError(f"rope_2d: unsupported dtype {tokens.dtype}")
    positions = positions.to(torch.int64).contiguous()
    _launch(tokens, positions, B * H, N, 1, D, N * D, D, D, H, base, fwd)


class cuRoPE2D_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, positions, base, F0=1):
        ctx.save_for_backward(positions)
        ctx.saved_base = base
        ctx.saved_F0 = F0
        rope_2d(tokens, positions, base, F0)
        ctx.mark_dirty(tokens)
        return tokens

    @staticmethod
    def backward(ctx, grad_res):
        positions, base, F0 = ctx.saved_tensors[0], ctx.saved_base, ctx.saved_F0
        if not (grad_res.stride(3) == 1 and grad_res.stride(2) == grad_res.size(3)):
            grad_res = grad_res.contiguous()
        rope_2d(grad_res, positions, base, -F0)
        ctx.mark_dirty(grad_res)
        return grad_res, None, None, None


class cuRoPE2D(torch.nn.Module):
    def __init__(self, freq: float = 100.0, F0: float = 1.0):
        super().__init__()
        self.base = freq
        self.F0 = F0

    def forward(self, tokens, positions):
        """tokens [B,H,N,D] -> same tensor object, rotated in place through its (B,N,H,D) view."""
        cuRoPE2D_func.apply(tokens.transpose(1, 2), positions, self.base, self.F0)
        return tokens


RoPE2D = cuRoPE2D


class PositionGetter:
    """Producer of the ``positions`` operand: the (y, x) patch-grid coordinates of an h x w token grid, int64,
    contiguous, one private copy per call (croco/blocks.py:207-219; consumed by PatchEmbedDust3R.forward,
    croco/patch_embed.py:19-29).  Grids are cached per (h, w, device)."""

    def __init__(self):
        self.cache_positions = {}

    def __call__(self, b: int, h: int, w: int, device) -> torch.Tensor:
        key = (h, w, str(device))
        grid = self.cache_positions.get(key)
        if grid is None:
            n = torch.arange(h * w, dtype=torch.int64, device=device)
            grid = torch.stack((n // w, n % w), dim=-1)                      # row-major: y slow, x fast
            self.cache_positions[key] = grid
        return grid.unsqueeze(0).repeat(b, 1, 1)


def append_token_position(pos: torch.Tensor) -> torch.Tensor:
    """Position of one extra (intrinsics / pose) token appended after the patch tokens: (first token's y + last token's
    y + 1, first token's x) -- i.e. one row below the grid, column 0 (backbone_masked_croco.py:163-172, 192-201).
    ``pos`` [..., N, 2] -> [..., N+1, 2]."""
    extra = pos[..., 0:1, :].clone()
    extra[..., 0] += pos[..., -1:, 0] + 1
    return torch.cat((pos, extra), dim=-2)


class _RoPE2DHeadMajor(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, positions, base):
        out = tokens.contiguous().clone()
        rope_2d_head_major(out, positions, base, 1.0)
        ctx.save_for_backward(positions)
        ctx.base = base
        return out

    @staticmethod
    def backward(ctx, grad):
        g = grad.contiguous().clone()
        rope_2d_head_major(g, ctx.saved_tensors[0], ctx.base, -1.0)
        return g, None, None


class RotaryPositionEmbedding2D(torch.nn.Module):
    """Drop-in for VGGT's ``RotaryPositionEmbedding2D``
    (/root/reference/src/model/encoder/backbone/vggt/layers/rope.py:62-188): out-of-place, tokens [B,H,N,D],
    positions [B,N,2] (y, x).  Same rotation as curope (first half of D by y, second by x, pairs (t[q], t[q+D/4]),
    angle = pos / frequency^(q/(D/4))); the reference's version costs two embedding gathers, two concatenations
    and a host sync (``int(positions.max())``, rope.py:174) per call -- here it is one kernel.  ``scaling_factor`` is
    accepted and unused, as in the reference."""

    def __init__(self, frequency: float = 100.0, scaling_factor: float = 1.0):
        super().__init__()
        self.base_frequency = frequency
        self.scaling_factor = scaling_factor

    def forward(self, tokens: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        assert tokens.size(-1) % 2 == 0, "Feature dimension must be even"
        assert positions.ndim == 3 and positions.shape[-1] == 2, "Positions must have shape (batch_size, n_tokens, 2)"
        return _RoPE2DHeadMajor.apply(tokens, positions, float(self.base_frequency))
