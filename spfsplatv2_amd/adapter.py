"""Fused Gaussian adapter on the HIP library: host mirror of the reference's ``UnifiedGaussianAdapter``
(/root/reference/src/model/encoder/common/gaussian_adapter.py:26-150).  One launch pair turns the network's raw
``7 + 3*d_sh`` channels into the scales / rotations / harmonics the decoder consumes (forward and backward), instead of
a chain of ~10 elementwise torch kernels.  ``covariances`` -- which no consumer of the decoder path reads
(cuda_splatting.py:136) -- are only materialised on request."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch
from torch import Tensor, nn

from . import _lib


@dataclass
class Gaussians:                       # field order of the adapter's own dataclass (gaussian_adapter.py:15-22)
    means: Tensor
    covariances: Tensor
    scales: Tensor
    rotations: Tensor
    harmonics: Tensor
    opacities: Tensor


@dataclass
class GaussianAdapterCfg:
    gaussian_scale_min: float
    gaussian_scale_max: float
    sh_degree: int


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class _AdapterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, mask, eps):
        ctx.set_materialize_grads(False)
        lib = _lib.load()
        N, Cn = raw.shape
        K = (Cn - 7) // 3
        scales = torch.empty((N, 3), dtype=torch.float32, device=raw.device)
        rot = torch.empty((N, 4), dtype=torch.float32, device=raw.device)
        sh = torch.empty((N, 3, K), dtype=torch.float32, device=raw.device)
        with torch.cuda.device(raw.device):
            _lib.check(lib.spf_adapter_forward(_p(raw), N, K, _p(mask), float(eps), _p(scales), _p(rot), _p(sh),
                                               C.c_void_p(torch.cuda.current_stream(raw.device).cuda_stream)),
                       "spf_adapter_forward")
        ctx.save_for_backward(raw, mask)
        ctx.eps = float(eps)
        return scales, rot, sh

    @staticmethod
    def backward(ctx, g_scales, g_rot, g_sh):
        lib = _lib.load()
        raw, mask = ctx.saved_tensors
        N, Cn = raw.shape
        K = (Cn - 7) // 3
        c = lambda g: None if g is None else g.contiguous().float()
        g_scales, g_rot, g_sh = c(g_scales), c(g_rot), c(g_sh)
        g_raw = torch.empty_like(raw)
        with torch.cuda.device(raw.device):
            _lib.check(lib.spf_adapter_backward(_p(raw), N, K, _p(mask), ctx.eps, _p(g_scales), _p(g_rot), _p(g_sh),
                                                _p(g_raw),
                                                C.c_void_p(torch.cuda.current_stream(raw.device).cuda_stream)),
                       "spf_adapter_backward")
        return g_raw, None, None


class UnifiedGaussianAdapter(nn.Module):
    """``forward(means, opacities, raw_gaussians, eps=1e-8) -> Gaussians`` with the reference's semantics."""

    def __init__(self, cfg: GaussianAdapterCfg):
        super().__init__()
        self.cfg = cfg
        mask = torch.ones((self.d_sh,), dtype=torch.float32)
        for degree in range(1, cfg.sh_degree + 1):
            mask[degree ** 2:(degree + 1) ** 2] = 0.1 * 0.25 ** degree
        self.register_buffer("sh_mask", mask, persistent=False)

    @property
    def d_sh(self) -> int:
        return (self.cfg.sh_degree + 1) ** 2

    @property
    def d_in(self) -> int:
        return 7 + 3 * self.d_sh

    def forward(self, means: Tensor, opacities: Tensor, raw_gaussians: Tensor, eps: float = 1e-8,
                with_covariances: bool = True) -> Gaussians:
        """``with_covariances`` (default True, as the reference: gaussian_adapter.py:140-141 always builds them, and
        encoder_spfsplatv2.py:302, validation_in_3d.py and validate_in_the_wild.py read them).  The render path never
        does (cuda_splatting.py:136), so a training loop may pass False to skip the 36 bytes per Gaussian: the field is
        then a zero-stride NaN tensor -- any accidental use is loud, never a silent zero."""
        if raw_gaussians.shape[-1] != self.d_in:
            raise RuntimeError(f"raw_gaussians has {raw_gaussians.shape[-1]} channels, expected {self.d_in}")
        if not raw_gaussians.is_cuda:
            raise RuntimeError("UnifiedGaussianAdapter: tensors are on the CPU; this build only runs on a HIP device")
        batch = raw_gaussians.shape[:-1]
        raw = raw_gaussians.reshape(-1, self.d_in).contiguous().float()
        scales, rot, sh = _AdapterFn.apply(raw, self.sh_mask.to(raw.device), eps)
        scales, rot = scales.reshape(*batch, 3), rot.reshape(*batch, 4)
        sh = sh.reshape(*batch, 3, self.d_sh).broadcast_to((*opacities.shape, 3, self.d_sh))
        if with_covariances:
            from .adapter_cov import build_covariance
            cov = build_covariance(scales, rot)
        else:
            cov = torch.full((), float("nan"), dtype=torch.float32, device=raw.device).expand(*batch, 3, 3)
        return Gaussians(means=means, covariances=cov, scales=scales,
                         rotations=rot.broadcast_to((*scales.shape[:-1], 4)), harmonics=sh, opacities=opacities)
