"""Photometric MSE loss on the HIP library: host mirror of the reference's ``LossMse``.

=====================  =========================================================================
here                   reference (/root/reference/src/loss/)
=====================  =========================================================================
``LossMseCfg``         loss_mse.py:13-16
``LossMseCfgWrapper``  loss_mse.py:18-20
``Loss``               loss.py:17-40 (the config wrapper convention: one dataclass field = the loss's name)
``LossMse``            loss_mse.py:36-51: ``weight * ((prediction - image) ** 2).mean()``, 0 before ``apply_after_step``
``mse_loss``           the same expression as a function (what ``bench.py`` and the tests call)
=====================  =========================================================================

The reference evaluates the expression with eager PyTorch (four kernels forward, four backward over the rendered
batch); here it is one pass forward and one backward, and the sum is taken in a fixed order (bit-reproducible).
No CPU path: tensors must live on a HIP device.
"""
from __future__ import annotations

import ctypes as C
from abc import ABC, abstractmethod
from dataclasses import dataclass, fields
from typing import Generic, TypeVar

import torch
from torch import Tensor, nn

from . import _lib

T_cfg = TypeVar("T_cfg")
T_wrapper = TypeVar("T_wrapper")


def _check(prediction: Tensor, image: Tensor) -> None:
    for name, t in (("prediction", prediction), ("image", image)):
        if not t.is_cuda:
            raise RuntimeError(f"mse_loss: {name} is on {t.device}; this build only runs on a HIP device (no CPU "
                               "fallback)")
        if not t.is_floating_point():
            raise RuntimeError(f"mse_loss: {name} must be a floating-point tensor, got {t.dtype}")
    if prediction.numel() == 0 or image.numel() == 0:
        raise RuntimeError("mse_loss: empty input")


def _kernel_operand(t: Tensor) -> Tensor:
    """What the kernel reads: float32, contiguous, 16-byte aligned (a slice of an odd-sized image is contiguous but
    may start anywhere: copied once rather than refused)."""
    t = t.to(torch.float32).contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone(memory_format=torch.contiguous_format)


_ONES: dict = {}


def unit_grad(device) -> Tensor:
    """THE dL/dloss = 1 of this process on `device`: a cached 0-dim float32 one.  `loss.backward(gradient=unit_grad(dev))`
    saves autograd's fill kernel, and the loss's backward recognises this very tensor (by its storage) and hands the
    forward's unit gradient on without even the one-scalar look of `spf_mse_scale_grad`: no launch at all.  Never write
    to it."""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device(dev.type, torch.cuda.current_device())
    t = _ONES.get(dev)
    if t is None:
        t = _ONES[dev] = torch.ones((), dtype=torch.float32, device=dev)
    return t


class _Mse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prediction: Tensor, image: Tensor, weight: float, grad_mode: bool = True):
        p, t = _kernel_operand(prediction), _kernel_operand(image)
        lib = _lib.load()
        dev = p.device
        # per-call scratch for the block partials (a few KB from the caching allocator): two streams computing
        # losses at the same time never share it
        partial = torch.empty(lib.spf_mse_partial_blocks(), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        # when a backward will come, the forward pass also writes the gradient for dL/dloss = 1 (what
        # `loss.backward()` passes): the backward is then one scalar look at the upstream gradient
        # (`grad_mode` = torch.is_grad_enabled() at the CALL SITE: needs_input_grad stays True under no_grad, and an
        #  evaluation loss on a tensor that requires grad must not pay a batch-sized allocation and write)
        unit = torch.empty_like(p) if (grad_mode and ctx.needs_input_grad[0]) else None
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            if unit is not None:
                _lib.check(lib.spf_mse_forward_grad(C.c_void_p(p.data_ptr()), C.c_void_p(t.data_ptr()), p.numel(),
                                                    float(weight), C.c_void_p(partial.data_ptr()),
                                                    C.c_void_p(loss.data_ptr()), C.c_void_p(unit.data_ptr()), stream),
                           "spf_mse_forward_grad")
            else:
                _lib.check(lib.spf_mse_forward(C.c_void_p(p.data_ptr()), C.c_void_p(t.data_ptr()), p.numel(),
                                               float(weight), C.c_void_p(partial.data_ptr()),
                                               C.c_void_p(loss.data_ptr()), stream),
                           "spf_mse_forward")
        ctx.save_for_backward(p, t)
        ctx.unit = unit
        ctx.weight = float(weight)
        ctx.shape = prediction.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        p, t = ctx.saved_tensors
        lib = _lib.load()
        dev = p.device
        g = g.to(torch.float32).contiguous()
        unit, ctx.unit = ctx.unit, None
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            if unit is not None:
                # first backward through this node: the forward's unit gradient, scaled in place by dL/dloss (a no-op
                # launch when that is 1).  The buffer is handed to autograd; a second backward (retain_graph) recomputes.
                gp = unit
                one = _ONES.get(g.device)
                if not (one is not None and g.data_ptr() == one.data_ptr()):     # (unit_grad(): exactly 1, nothing to do)
                    _lib.check(lib.spf_mse_scale_grad(C.c_void_p(gp.data_ptr()), gp.numel(), C.c_void_p(g.data_ptr()),
                                                      stream), "spf_mse_scale_grad")
            else:
                gp = torch.empty_like(p)
                _lib.check(lib.spf_mse_backward(C.c_void_p(p.data_ptr()), C.c_void_p(t.data_ptr()), p.numel(),
                                                ctx.weight, C.c_void_p(g.data_ptr()), C.c_void_p(gp.data_ptr()), stream),
                           "spf_mse_backward")
        gp = gp.view(ctx.shape)
        gi = -gp if ctx.needs_input_grad[1] else None
        return gp, gi, None, None


def mse_loss(prediction: Tensor, image: Tensor, weight: float = 1.0) -> Tensor:
    """``weight * ((prediction - image) ** 2).mean()`` (loss_mse.py:48-51) as one fused pass; 0-dim float32 result.
    Like the reference's expression it accepts broadcastable shapes and any floating dtype (bf16 under autocast):
    operands are expanded / cast to float32 on the host side of the kernel and the gradients are cast / reduced back
    by autograd."""
    _check(prediction, image)
    if prediction.shape != image.shape:
        prediction, image = torch.broadcast_tensors(prediction, image)
    if prediction.dtype != torch.float32:
        prediction = prediction.float()
    if image.dtype != torch.float32:
        image = image.float()
    return _Mse.apply(prediction, image, weight, torch.is_grad_enabled())


class Loss(nn.Module, ABC, Generic[T_cfg, T_wrapper]):
    cfg: T_cfg
    name: str

    def __init__(self, cfg: T_wrapper) -> None:
        super().__init__()
        (field,) = fields(type(cfg))            # the wrapper's single field names the loss (loss.py:24-31)
        self.cfg = getattr(cfg, field.name)
        self.name = field.name

    @abstractmethod
    def forward(self, prediction, batch, gaussians, global_step: int) -> Tensor:
        ...


@dataclass
class LossMseCfg:
    weight: float
    apply_after_step: int


@dataclass
class LossMseCfgWrapper:
    mse: LossMseCfg


class LossMse(Loss[LossMseCfg, LossMseCfgWrapper]):
    def forward(self, prediction: Tensor, image: Tensor, gaussians, global_step: int) -> Tensor:
        if global_step < self.cfg.apply_after_step:           # not applied yet (loss_mse.py:44-46)
            return torch.tensor(0, dtype=torch.float32, device=image.device)
        return mse_loss(prediction, image, self.cfg.weight)
