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


def rope_2d(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> None:
    """In-place RoPE-2D on ``tokens`` (B,N,H,D); ``fwd`` = +F0 forward, -F0 backward."""
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
    B, N, H, D = tokens.shape
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
    lib = _lib.load()
    stream = C.c_void_p(torch.cuda.current_stream(tokens.device).cuda_stream)
    _lib.check(lib.spf_rope2d(C.c_void_p(tokens.data_ptr()), C.c_void_p(positions.data_ptr()), B, N, H, D,
                              tokens.stride(0), tokens.stride(1), _DTYPES[tokens.dtype], float(base), float(fwd),
                              stream), "spf_rope2d")


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
