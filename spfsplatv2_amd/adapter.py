"""Fused Gaussian adapter on the HIP library: host mirror of the reference's ``UnifiedGaussianAdapter``
(/root/reference/src/model/encoder/common/gaussian_adapter.py:26-150).  One launch pair turns the network's raw
``7 + 3*d_sh`` channels into the scales / rotations / harmonics the decoder consumes (forward and backward), instead of
a chain of ~10 elementwise torch kernels.  ``covariances`` -- which no consumer of the decoder path reads
(cuda_splatting.py:136) -- are only materialised on request."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

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
    harmonics_band4: Optional[Tensor] = None    # band-split harmonics: see decoder.Gaussians.harmonics_band4
    raw: Optional[RawGaussians] = None          # fused mode: scales / rotations / harmonics are None, see RawGaussians


@dataclass
class GaussianAdapterCfg:
    gaussian_scale_min: float
    gaussian_scale_max: float
    sh_degree: int


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


from .rasterizer import raw_rows as _rows          # (rows read in place: see there)


@dataclass
class RawGaussians:
    """The adapter FUSED INTO THE DECODER (SpfDims.sh_layout 3): the network's raw channels as they are, plus what the
    adapter would have applied to them.  `DecoderSplattingCUDA` consumes this directly -- the projection kernels form
    scales, rotations and masked harmonics as they read a row and chain the backward through them -- so the adapter's own
    pass over the tensor (656 bytes per Gaussian forward, 576 backward: more than the decoder itself moves) never runs."""
    raw: Tensor          # [..., 7 + 3*d_sh]: any leading shape, unit stride along the channels (a view of the head output)
    sh_mask: Tensor      # [d_sh]
    eps: float


class _AdapterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw_any, mask, eps, split, d_in):
        # (`raw_any`: the caller's tensor, whatever its shape; its [N, d_in] rows are taken HERE, outside autograd's view --
        #  as an autograd op the as_strided of a view costs a zero fill and a scatter copy of the whole tensor in backward)
        ctx.set_materialize_grads(False)
        lib = _lib.load()
        raw = _rows(raw_any.detach(), d_in)
        ctx.raw_shape = tuple(raw_any.shape)
        N, Cn = raw.shape
        K = (Cn - 7) // 3
        f32 = dict(dtype=torch.float32, device=raw.device)
        scales = torch.empty((N, 3), **f32)
        rot = torch.empty((N, 4), **f32)
        sh = torch.empty((N, 3, 16 if split else K), **f32)
        sh_hi = torch.empty((N, 3, 9), **f32) if split else None
        with torch.cuda.device(raw.device):
            _lib.check(lib.spf_adapter_forward(_p(raw), raw.stride(0), N, K, _p(mask), float(eps), _p(scales), _p(rot),
                                               _p(sh), _p(sh_hi),
                                               C.c_void_p(torch.cuda.current_stream(raw.device).cuda_stream)),
                       "spf_adapter_forward")
        ctx.save_for_backward(raw, mask)
        ctx.eps, ctx.split = float(eps), bool(split)
        return (scales, rot, sh, sh_hi) if split else (scales, rot, sh)

    @staticmethod
    def backward(ctx, g_scales, g_rot, g_sh, g_sh_hi=None):
        lib = _lib.load()
        raw, mask = ctx.saved_tensors
        N, Cn = raw.shape
        K = (Cn - 7) // 3
        c = lambda g: None if g is None else g.contiguous().float()
        # (band split: a decoder that evaluates to degree 3 hands back NO gradient for the band-4 plane -- None here --
        #  and the kernel then writes zeros for those channels without reading anything)
        g_scales, g_rot, g_sh, g_sh_hi = c(g_scales), c(g_rot), c(g_sh), c(g_sh_hi)
        g_raw = torch.empty((N, Cn), dtype=torch.float32, device=raw.device)
        with torch.cuda.device(raw.device):
            _lib.check(lib.spf_adapter_backward(_p(raw), raw.stride(0), N, K, _p(mask), ctx.eps, _p(g_scales), _p(g_rot),
                                                _p(g_sh), _p(g_sh_hi), 1 if ctx.split else 0, _p(g_raw),
                                                C.c_void_p(torch.cuda.current_stream(raw.device).cuda_stream)),
                       "spf_adapter_backward")
        return g_raw.view(ctx.raw_shape), None, None, None, None


class UnifiedGaussianAdapter(nn.Module):
    """``forward(means, opacities, raw_gaussians, eps=1e-8) -> Gaussians`` with the reference's semantics."""

    def __init__(self, cfg: GaussianAdapterCfg, split_harmonics: bool = False, fuse_into_decoder: bool = False):
        """``split_harmonics`` (d_sh = 25 only; not an argument of the reference's class): write the harmonics BAND-SPLIT,
        ``Gaussians.harmonics`` = bands 0 - 3 [.., 3, 16] and ``Gaussians.harmonics_band4`` = band 4 [.., 3, 9] -- the
        layout in which the decoder's default evaluation depth (degree 3) leaves band 4's third of every
        coefficient block in HBM, forward and backward (SpfDims.sh_layout 2)."""
        super().__init__()
        self.cfg = cfg
        # `fuse_into_decoder` (not an argument of the reference's class): `forward` launches NOTHING -- it returns a
        # `Gaussians` whose `raw` field carries the raw channels, the SH mask and eps, and whose scales / rotations /
        # harmonics are None (`materialize()` produces them for any other consumer); the decoder applies the adapter inside
        # its projection kernels.  Same images and gradients, to float32 rounding (tests/test_gpu_adapter.py).
        self.fuse_into_decoder = bool(fuse_into_decoder)
        if split_harmonics and cfg.sh_degree != 4:
            raise ValueError("split_harmonics is the 16 + 9 split of sh_degree 4 (d_sh = 25)")
        self.split_harmonics = bool(split_harmonics)
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
        if self.fuse_into_decoder:
            return Gaussians(means=means, covariances=None, scales=None, rotations=None, harmonics=None,
                             opacities=opacities, raw=RawGaussians(raw_gaussians, self.sh_mask.to(raw_gaussians.device),
                                                                   float(eps)))
        batch = raw_gaussians.shape[:-1]
        raw = raw_gaussians
        sh_hi = None
        if self.split_harmonics:
            scales, rot, sh, sh_hi = _AdapterFn.apply(raw, self.sh_mask.to(raw.device), eps, True, self.d_in)
            sh_hi = sh_hi.reshape(*batch, 3, 9).broadcast_to((*opacities.shape, 3, 9))
        else:
            scales, rot, sh = _AdapterFn.apply(raw, self.sh_mask.to(raw.device), eps, False, self.d_in)
        scales, rot = scales.reshape(*batch, 3), rot.reshape(*batch, 4)
        sh = sh.reshape(*batch, 3, sh.shape[-1]).broadcast_to((*opacities.shape, 3, sh.shape[-1]))
        if with_covariances:
            from .adapter_cov import build_covariance
            cov = build_covariance(scales, rot)
        else:
            cov = torch.full((), float("nan"), dtype=torch.float32, device=raw.device).expand(*batch, 3, 3)
        return Gaussians(means=means, covariances=cov, scales=scales,
                         rotations=rot.broadcast_to((*scales.shape[:-1], 4)), harmonics=sh, opacities=opacities,
                         harmonics_band4=sh_hi)


def materialize(g: Gaussians, split_harmonics: bool = False) -> Gaussians:
    """The standard fields of a fused-mode `Gaussians` (scales, rotations, harmonics), through the adapter kernels -- for
    the consumers that are not the decoder (visualisation dumps, export_ply)."""
    if g.raw is None:
        return g
    raw = g.raw.raw
    K = (raw.shape[-1] - 7) // 3
    batch = raw.shape[:-1]
    if split_harmonics:
        scales, rot, sh, hi = _AdapterFn.apply(raw, g.raw.sh_mask, g.raw.eps, True, raw.shape[-1])
        hi = hi.reshape(*batch, 3, 9)
    else:
        scales, rot, sh = _AdapterFn.apply(raw, g.raw.sh_mask, g.raw.eps, False, raw.shape[-1])
        hi = None
    return Gaussians(means=g.means, covariances=g.covariances, scales=scales.reshape(*batch, 3),
                     rotations=rot.reshape(*batch, 4), harmonics=sh.reshape(*batch, 3, sh.shape[-1]), opacities=g.opacities,
                     harmonics_band4=hi)
