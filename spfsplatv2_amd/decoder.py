"""Decoder glue: the reference's ``src/model/decoder`` package re-stated on top of the batched HIP
rasterizer.  Same names, argument meaning and error behaviour as the reference so that callers
(`ModelWrapper.training_step` etc.) can switch by changing one import:

=============================  =====================================================================
here                           reference
=============================  =====================================================================
``Gaussians``                  src/model/types.py:7-14
``DecoderOutput``, ``Decoder`` src/model/decoder/decoder.py:18-45
``get_projection_matrix``      src/model/decoder/cuda_splatting.py:15-42
``get_fov``                    src/geometry/projection.py:269-283
``render_cuda``                src/model/decoder/cuda_splatting.py:45-144
``render_cuda_orthographic``   src/model/decoder/cuda_splatting.py:146-255
``DecoderSplattingCUDACfg``    src/model/decoder/decoder_splatting_cuda.py:15-21
``DecoderSplattingCUDA``       src/model/decoder/decoder_splatting_cuda.py:23-78
``DECODERS``, ``get_decoder``  src/model/decoder/__init__.py:4-12
=============================  =====================================================================

What changes underneath (results are the same):
* no Python loop over views and no ``.item()`` host syncs (cuda_splatting.py:96-143,108-109): one
  batched launch chain, tan(fov/2) stays on the device;
* no ``repeat`` of every Gaussian tensor per view (decoder_splatting_cuda.py:59-64): the V views of a
  scene share its Gaussian buffers, and the scale-invariant normalisation (cuda_splatting.py:66-74)
  is applied per view inside the projection kernel (``view_scale``);
* the SH block is handed over in the layout the kernel reads, ``[.., K, 3]`` (cuda_splatting.py:79).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from math import isqrt
from typing import Generic, Literal, Optional, TypeVar

import torch
from torch import Tensor, nn

import os

from ._lib import SpfError as _SpfError
from .rasterizer import CallRecord, PairBudget, rasterize_batch, render_batch, sh_band4_default

DepthRenderingMode = Literal["depth", "log", "disparity", "relative_disparity"]


def _auto_plan_from_env() -> Optional[float]:
    """SPF_AUTO_PLAN: slack factor of the module's own planning (default 1.5; 0 / empty: off).  A typo is an error that
    names the variable, not a bare float() failure inside a constructor."""
    raw = os.environ.get("SPF_AUTO_PLAN", "1.5").strip()
    if not raw:
        return None
    try:
        v = float(raw)
    except ValueError:
        raise ValueError(f"SPF_AUTO_PLAN={raw!r} is not a number (slack factor >= 1, or 0 for off)") from None
    if v != 0 and not v >= 1.0:
        raise ValueError(f"SPF_AUTO_PLAN={raw!r}: the slack factor must be >= 1 (or 0 for off)")
    return v or None


@dataclass
class Gaussians:
    means: Tensor        # [b, g, 3]
    covariances: Tensor  # [b, g, 3, 3]   (carried, never read by the rasterizer: cuda_splatting.py:136)
    rotations: Tensor    # [b, g, 4]
    scales: Tensor       # [b, g, 3]
    harmonics: Tensor    # [b, g, 3, d_sh]   ([b, g, 3, 16] when `harmonics_band4` is given)
    opacities: Tensor    # [b, g]
    # (not a field of the reference's dataclass) BAND-SPLIT harmonics of a d_sh = 25 model: `harmonics` holds bands 0 - 3
    # and this [b, g, 3, 9] tensor band 4 -- what `UnifiedGaussianAdapter(..., split_harmonics=True)` produces.  The
    # default evaluation depth (degree 3) then never moves band 4's bytes; None: `harmonics` is the reference's layout.
    harmonics_band4: Optional[Tensor] = None
    # (not a field of the reference's dataclass) the adapter FUSED INTO THE DECODER: an `adapter.RawGaussians` (raw network
    # channels [b, g, 7 + 3 d_sh] + SH mask + eps, from `UnifiedGaussianAdapter(..., fuse_into_decoder=True)`); scales,
    # rotations and harmonics are then None and the projection kernels apply the adapter as they read a row.
    raw: Optional[object] = None


@dataclass
class DecoderOutput:
    color: Tensor            # [b, v, 3, h, w]
    depth: Optional[Tensor]  # [b, v, h, w]


def get_projection_matrix(near: Tensor, far: Tensor, fov_x: Tensor, fov_y: Tensor) -> Tensor:
    """Perspective matrix [B,4,4]: x, y -> (-1, 1), z -> (0, 1), w = z_view (column-vector form)."""
    tan_x = (0.5 * fov_x).tan()
    tan_y = (0.5 * fov_y).tan()
    top, right = tan_y * near, tan_x * near
    bottom, left = -top, -right
    (b,) = near.shape
    out = torch.zeros((b, 4, 4), dtype=torch.float32, device=near.device)
    out[:, 0, 0] = 2 * near / (right - left)
    out[:, 1, 1] = 2 * near / (top - bottom)
    out[:, 0, 2] = (right + left) / (right - left)
    out[:, 1, 2] = (top + bottom) / (top - bottom)
    out[:, 3, 2] = 1
    out[:, 2, 2] = far / (far - near)
    out[:, 2, 3] = -(far * near) / (far - near)
    return out


def get_fov(intrinsics: Tensor) -> Tensor:
    """Field of view [B,2] (x, y) in radians from normalised intrinsics [B,3,3]: angle between the
    unit rays through the midpoints of opposite image edges."""
    inv = intrinsics.inverse()

    def ray(u: float, v: float) -> Tensor:
        p = torch.tensor([u, v, 1.0], dtype=torch.float32, device=intrinsics.device)
        d = (inv * p[None, None, :]).sum(dim=-1)
        return d / d.norm(dim=-1, keepdim=True)

    fov_x = (ray(0.0, 0.5) * ray(1.0, 0.5)).sum(dim=-1).acos()
    fov_y = (ray(0.5, 0.0) * ray(0.5, 1.0)).sum(dim=-1).acos()
    return torch.stack((fov_x, fov_y), dim=-1)


def _camera_tensors(extrinsics: Tensor, near: Tensor, far: Tensor, fov_x: Tensor, fov_y: Tensor):
    """Row-vector view / projection matrices (the transposes built at cuda_splatting.py:88-90)."""
    proj = get_projection_matrix(near, far, fov_x, fov_y).transpose(-1, -2)
    view = extrinsics.inverse().transpose(-1, -2)
    return view, proj


def render_views(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, image_shape: tuple[int, int],
                 background_color: Tensor, gaussian_means: Tensor, gaussian_sh_coefficients: Tensor,
                 gaussian_opacities: Tensor, gaussian_rotations: Tensor, gaussian_scales: Tensor,
                 scale_invariant: bool = True, use_sh: bool = True, enable_cov_grad: bool = False,
                 enable_sh_grad: bool = False, max_pairs=None, sh_band4: Optional[bool] = None,
                 return_radii: bool = False, record=None, gaussian_sh_band4: Optional[Tensor] = None,
                 gaussian_raw=None):
    """Batched form of ``render_cuda``: b scenes x v views sharing each scene's Gaussians.

    extrinsics [b,v,4,4] (camera-to-world), intrinsics [b,v,3,3] (normalised), near/far [b,v],
    background_color [3] or [b,v,3], gaussian_* [b,g,...] with SH as [b,g,3,d_sh].
    Returns color [b,v,3,h,w], depth [b,v,1,h,w] (in the rasterizer's normalised units: the caller
    multiplies by near, decoder_splatting_cuda.py:72-76) and alpha [b,v,1,h,w] (+ radii [b,v,g] int32 with
    ``return_radii``).  ``sh_band4``: with d_sh = 25 (sh_degree 4, the reference's default) also evaluate SH band 4;
    None = the ``SPF_SH_BAND4`` environment variable, default off (see ``rasterizer.sh_band4_default``).
    ``gaussian_sh_band4`` [b,g,3,9]: band 4 of BAND-SPLIT harmonics, ``gaussian_sh_coefficients`` then being [b,g,3,16]
    (``Gaussians.harmonics_band4``).
    """
    h, w = image_shape
    if gaussian_raw is not None:
        # the adapter fused into the projection kernels (`Gaussians.raw`): scales / rotations / harmonics are not looked at
        n = (gaussian_raw.raw.shape[-1] - 7) // 3
        color, depth, alpha, _radii = render_batch(
            extrinsics, intrinsics, near, far, gaussian_means, None, None, gaussian_opacities, None, None,
            background_color, h, w, isqrt(n) - 1, scale_invariant, True, True, max_pairs=max_pairs, sh_layout="g3k",
            sh_band4=sh_band4, record=record, raw=gaussian_raw.raw, sh_mask=gaussian_raw.sh_mask,
            adapter_eps=gaussian_raw.eps)
        return (color, depth, alpha, _radii) if return_radii else (color, depth, alpha)
    assert use_sh or gaussian_sh_coefficients.shape[-1] == 1
    n = gaussian_sh_coefficients.shape[-1] + (0 if gaussian_sh_band4 is None else gaussian_sh_band4.shape[-1])
    degree = isqrt(n) - 1
    # SH stays in the encoder's [b,g,3,d_sh] layout: the kernels index it directly (no transposed copy as at
    # cuda_splatting.py:79); colours-only mode takes the DC coefficient as the colour (cuda_splatting.py:132)
    color, depth, alpha, _radii = render_batch(
        extrinsics, intrinsics, near, far, gaussian_means, gaussian_scales, gaussian_rotations, gaussian_opacities,
        gaussian_sh_coefficients if use_sh else None, None if use_sh else gaussian_sh_coefficients[..., 0],
        background_color, h, w, degree, scale_invariant, enable_cov_grad, enable_sh_grad, max_pairs=max_pairs,
        sh_layout="g3k", sh_band4=sh_band4, record=record, shs_high=gaussian_sh_band4 if use_sh else None)
    return (color, depth, alpha, _radii) if return_radii else (color, depth, alpha)


def camera_tensors(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, scale_invariant: bool = True):
    """The camera preamble of ``render_cuda`` (cuda_splatting.py:66-74,84-91) as plain torch ops, for callers
    that drive ``rasterize_batch`` / ``GaussianRasterizer`` themselves: flat batch [B,...] in, row-vector
    view / projection matrices [B,4,4], tanfov [B,2] and the per-render world scale [B] out.  (The decoder itself
    uses the fused HIP camera kernel through ``render_batch``.)"""
    scale = torch.ones_like(near)
    if scale_invariant:
        scale = 1 / near
        extrinsics = extrinsics.clone()
        extrinsics[..., :3, 3] = extrinsics[..., :3, 3] * scale[..., None]
        far = far * scale
        near = near * scale
    fov_x, fov_y = get_fov(intrinsics).unbind(dim=-1)
    tanfov = torch.stack(((0.5 * fov_x).tan(), (0.5 * fov_y).tan()), dim=-1)
    view, proj = _camera_tensors(extrinsics, near, far, fov_x, fov_y)
    return view, proj, tanfov, scale


def render_cuda(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, image_shape: tuple[int, int],
                background_color: Tensor, gaussian_means: Tensor, gaussian_covariances: Tensor,
                gaussian_sh_coefficients: Tensor, gaussian_opacities: Tensor, gaussian_rotations: Tensor,
                gaussian_scales: Tensor, scale_invariant: bool = True, use_sh: bool = True,
                enable_cov_grad: bool = False, enable_sh_grad: bool = False):
    """Same signature and result as the reference's ``render_cuda`` (flat batch: item i renders its own
    Gaussian set ``gaussian_*[i]`` from camera i).  Returns (images [B,3,h,w], depths [B,1,h,w])."""
    del gaussian_covariances  # dead in the reference too (cuda_splatting.py:136)
    color, depth, _ = render_views(
        extrinsics[:, None], intrinsics[:, None], near[:, None], far[:, None], image_shape,
        background_color[:, None], gaussian_means, gaussian_sh_coefficients, gaussian_opacities,
        gaussian_rotations, gaussian_scales, scale_invariant, use_sh, enable_cov_grad, enable_sh_grad)
    return color[:, 0], depth[:, 0]


def render_cuda_orthographic(extrinsics: Tensor, width: Tensor, height: Tensor, near: Tensor, far: Tensor,
                             image_shape: tuple[int, int], background_color: Tensor, gaussian_means: Tensor,
                             gaussian_covariances: Tensor, gaussian_sh_coefficients: Tensor,
                             gaussian_opacities: Tensor, gaussian_rotations: Tensor, gaussian_scales: Tensor,
                             fov_degrees: float = 0.1, use_sh: bool = True, dump: dict | None = None,
                             enable_cov_grad: bool = False, enable_sh_grad: bool = False) -> Tensor:
    """Fake orthographic render (tiny FOV, camera moved back); returns images [B,3,h,w]."""
    del gaussian_covariances
    h, w = image_shape
    assert use_sh or gaussian_sh_coefficients.shape[-1] == 1
    n = gaussian_sh_coefficients.shape[-1]
    degree = isqrt(n) - 1
    shs = gaussian_sh_coefficients.transpose(-1, -2).contiguous()
    view, proj, tanfov = orthographic_camera(extrinsics, width, height, near, far, fov_degrees, dump)
    color, _, _, _ = rasterize_batch(
        gaussian_means, gaussian_scales, gaussian_rotations, gaussian_opacities,
        shs if use_sh else None, None if use_sh else shs[:, :, 0, :],
        view[:, None], proj[:, None], tanfov, background_color[:, None],
        h, w, degree, 1.0, enable_cov_grad, enable_sh_grad)
    return color[:, 0]


def orthographic_camera(extrinsics: Tensor, width: Tensor, height: Tensor, near: Tensor, far: Tensor,
                        fov_degrees: float = 0.1, dump: dict | None = None):
    """Camera tensors of ``render_cuda_orthographic`` (cuda_splatting.py:168-201): view / projection matrices [B,4,4]
    (row-vector form) and tanfov [B,1,2] of the tiny-FOV camera moved back by ``distance_to_near``.

    The reference inverts the moved-back pose, ``(extrinsics @ move_back).inverse()``: a float32 inverse of a matrix
    whose translation is ~0.5*width/tan(fov/2) ~ 10^3 scene units, good to ~1e-7 of THAT -- pixel centres wobble by
    ~1e-3 px from one inverse routine to the next.  The same matrix is ``move_back^-1 @ extrinsics^-1``, i.e. the
    inverse of the ORIGINAL pose with ``distance_to_near`` added to its z translation -- identical in exact arithmetic
    and free of the cancellation, so that is what is built here."""
    b = extrinsics.shape[0]
    fov_x = torch.tensor(fov_degrees, device=extrinsics.device).deg2rad()
    tan_fov_x = (0.5 * fov_x).tan()
    distance_to_near = (0.5 * width) / tan_fov_x
    tan_fov_y = 0.5 * height / distance_to_near
    fov_y = (2 * tan_fov_y).atan()
    near = near + distance_to_near
    far = far + distance_to_near
    if dump is not None:
        # (the reference writes a single 4x4 here, cuda_splatting.py:183-185, which only works for batch 1)
        move_back = torch.eye(4, dtype=torch.float32, device=extrinsics.device).repeat(b, 1, 1)
        move_back[:, 2, 3] = -distance_to_near
        dump["extrinsics"] = extrinsics @ move_back
        dump["fov_x"] = fov_x
        dump["fov_y"] = fov_y
        dump["near"] = near
        dump["far"] = far
    w2c = extrinsics.inverse().clone()
    w2c[:, 2, 3] = w2c[:, 2, 3] + distance_to_near              # move_back^-1 @ w2c
    view = w2c.transpose(-1, -2)
    proj = get_projection_matrix(near, far, fov_x.expand(b), fov_y).transpose(-1, -2)
    tanfov = torch.stack((tan_fov_x.expand(b), tan_fov_y.expand(b)), dim=-1).reshape(b, 1, 2)
    return view, proj, tanfov


T = TypeVar("T")


class Decoder(nn.Module, ABC, Generic[T]):
    cfg: T

    def __init__(self, cfg: T) -> None:
        super().__init__()
        self.cfg = cfg

    @abstractmethod
    def forward(self, gaussians: Gaussians, extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                image_shape: tuple[int, int], depth_mode: DepthRenderingMode | None = None) -> DecoderOutput:
        pass


@dataclass
class DecoderSplattingCUDACfg:
    name: Literal["splatting_cuda"]
    background_color: list[float]
    make_scale_invariant: bool
    enable_cov_grad: bool
    enable_sh_grad: bool


_data_ptr = torch.Tensor.data_ptr


class _EvalGraph:
    """One captured evaluation call of a decoder: the graph, the tensors it writes (colour and depth packed into one
    flat buffer, alpha, radii -- owned by the graph's memory pool) and the call record whose `counters` it
    refreshes."""
    __slots__ = ("graph", "outputs", "record", "sizes", "color_shape", "depth_shape", "nbytes")

    def __init__(self, graph, outputs, record):
        self.graph, self.outputs, self.record = graph, outputs, record


class _PreparedStep:
    """One prepared TRAINING call of a decoder: the static step (state at fixed addresses, argument structs built once:
    rasterizer.StaticStep) and who holds it at the moment."""
    __slots__ = ("step", "record", "gen", "token", "nbytes", "__weakref__")

    def busy(self) -> bool:
        """A forward of this entry is still waiting for its backward (its state must not be overwritten)."""
        t = self.token() if self.token is not None else None
        return t is not None and not t.consumed


class _StepToken:
    """Lives as long as the autograd node of one prepared forward."""
    __slots__ = ("consumed", "gen", "__weakref__")

    def __init__(self, gen):
        self.consumed, self.gen = False, gen


class _PreparedRender(torch.autograd.Function):
    """A training call on a prepared step: ONE autograd node, like the eager path's; outputs and gradients are fresh
    tensors, the state in between is the entry's."""

    @staticmethod
    def forward(ctx, entry, check, want_extra, extrinsics, means, scales, rotations, opacities, shs, shs_high):
        import weakref

        from . import rasterizer as rz
        ctx.set_materialize_grads(False)
        step = entry.step
        entry.gen += 1
        token = _StepToken(entry.gen)
        entry.token = weakref.ref(token)
        ctx.entry, ctx.token, ctx.check = entry, token, check
        # (from the compiled step, csrc/torch_binding.cpp::PreparedStep, when it has been built: one C++ call)
        color, depth, alpha, failed = step.forward(check == "early")
        radii = step.radii.view(alpha.shape[0], alpha.shape[1], -1).clone() if want_extra else None
        if failed:
            token.consumed = True
            step.raise_if_failed()
        if want_extra:
            ctx.mark_non_differentiable(radii)
        return color, depth, (alpha if want_extra else None), radii

    @staticmethod
    def backward(ctx, g_color, g_depth, g_alpha, _g_radii):
        from . import rasterizer as rz
        from .shard import active_bucket
        entry, token = ctx.entry, ctx.token
        step = entry.step
        if token.gen != entry.gen:
            raise RuntimeError("spfsplatv2_amd: this decoder call's saved state was overwritten by a later call of the same "
                               "shapes (a SECOND backward through a prepared training call after a newer forward); "
                               "set decoder.prepare_steps = False for such a loop")
        if ctx.check == "backward" and not token.consumed:
            step.raise_if_failed()                   # (one host sync, as in the eager path: a failed plan raises here)
        token.consumed = True                        # (a retained graph's second backward finds the same state: gen matches)
        need = ctx.needs_input_grad      # (entry, check, want_extra, extrinsics, means, scales, rotations, opacities, shs, shs_high)
        if active_bucket() is None:
            if step.fast is not None:
                with rz._spf_errors():
                    d_means, d_opac, d_scales, d_rot, d_shs, d_high, d_ext = step.fast.backward(g_color, g_depth, g_alpha)
                return (None, None, None, d_ext if need[3] else None, d_means, d_scales, d_rot, d_opac, d_shs, d_high)
            with torch.cuda.device(step.dev):
                g = step.backward(g_color, g_depth, g_alpha)
            return (None, None, None, g.get("extrinsics") if need[3] else None, g["means"], g.get("scales"),
                    g.get("rotations"), g["opacities"], g.get("harmonics"), g.get("harmonics_band4"))
        step.ensure_python_binding()
        # a gradient bucket supplies the output buffers (shard.GradBucket): the general launcher, same state
        if g_depth is not None:
            g_depth = g_depth[:, :, None] * step.near_b if step.scale_invariant else g_depth[:, :, None]
        res = rz._backward_impl(step.inputs, step.state, step.geom, (g_color, g_depth, g_alpha),
                                dict(step.want, view="partials" if step.want["view"] else False), shs_high=step.shs_high)
        d_means, d_scales, d_rot, d_opac, d_shs, _d_col, vpartial, _ = res[:8]
        d_high = res[8] if len(res) > 8 else None
        d_ext = None
        if need[3] and vpartial is not None:
            import ctypes as C

            from . import _lib
            d_ext = torch.empty_like(step.view)
            with torch.cuda.device(step.dev):
                _lib.check(step.lib.spf_camera_backward_partials(C.byref(step.cam_b), rz._ptr(vpartial), vpartial.shape[1],
                                                                 rz._ptr(d_ext), rz._stream_ptr(step.dev)),
                           "spf_camera_backward_partials")
        return (None, None, None, d_ext, d_means, d_scales, d_rot, d_opac, d_shs, d_high)


class DecoderSplattingCUDA(Decoder[DecoderSplattingCUDACfg]):
    """Drop-in for the reference decoder (kept under the reference's registry name
    ``"splatting_cuda"``); the work runs on the MI355X HIP rasterizer."""

    background_color: Tensor

    def __init__(self, cfg: DecoderSplattingCUDACfg) -> None:
        super().__init__(cfg)
        self.make_scale_invariant = cfg.make_scale_invariant
        self.enable_cov_grad = cfg.enable_cov_grad
        self.enable_sh_grad = cfg.enable_sh_grad
        self.register_buffer("background_color", torch.tensor(cfg.background_color, dtype=torch.float32),
                             persistent=False)
        # `max_pairs` (property): None, or a ``spfsplatv2_amd.plan_pair_budget(...)`` the CALLER sets after one exact call --
        # then no call waits for the device any more (the plan is verified on the device, see rasterizer.PairBudget) and
        # the module's own planning (`auto_plan`, below) stands back.
        self._max_pairs = None
        self._auto_owned = False         # True while `_max_pairs` holds a plan the module made for itself
        self._auto_key = None            # shape the current automatic plan was made for
        self._auto_plans: dict = {}      # shape -> the plan it ran under last (a few shapes alternate in a real loop)
        self._auto_pending = None        # (deferred mode) (pinned verdict, event) of the last planned training call, until read
        # None: SH band 4 of a d_sh = 25 model follows the SPF_SH_BAND4 environment variable (default: not evaluated,
        # as in the published 3DGS kernels); True / False pins it for this decoder.
        self.sh_band4 = None
        # statistics / plan counters of THIS decoder's most recent call (two live decoders do not interleave):
        # `spfsplatv2_amd.plan_flags(decoder.last_call)`, `plan_pair_budget(decoder.last_call)`
        self.last_call = CallRecord()
        self._last_call_borrowed = False   # last_call is the record of a prepared step / captured graph (theirs to keep)
        # Evaluation-shaped calls (no gradient will be asked for, planned pair budget): the whole call -- camera set-up,
        # projection + binning, tile sort, compositing, depth x near -- is ~9 launches of a few microseconds each, and a
        # Python thread needs longer to issue them than the GPU to run them (test_step: b = 1, v = 3,
        # model_wrapper.py:415-454: 0.154 ms per call launched one by one, 0.093 ms as a replayed HIP graph).  So the
        # module keeps a small cache of captured graphs, keyed by EVERYTHING a graph bakes in: the addresses, shapes and
        # strides of all input tensors, the image size, the plan and the band-4 switch.  The first call of a key runs
        # as usual, the second is captured, later ones are one graph launch; results are copied out of the graph's own
        # buffers, so what a call returns is the caller's (never overwritten by a later call).  A loop that hands over
        # fresh addresses every time simply never hits (after `_EVAL_GRAPH_MISSES` captures that were never replayed the
        # cache stops capturing).  `eval_graphs = False` or SPF_EVAL_GRAPHS=0 switches it off; `clear_eval_graphs()`
        # releases the captured graphs and their buffers.  (Since the prepared steps below exist, evaluation calls of a
        # plan with direct bins run on those -- faster than a replay plus its copy-out, and indifferent to where the
        # tensors lie; this cache serves the plans that cannot be prepared and `prepare_steps = False`.)
        self.eval_graphs = True
        # `auto_plan` (slack factor, default 1.5; SPF_AUTO_PLAN=<slack>, SPF_AUTO_PLAN=0 / `auto_plan = None`: off): while the
        # caller has set no `max_pairs`, the module plans for itself -- an UNCHANGED caller does not pay exact mode's
        # read-back in the MIDDLE of every forward (project -> scan -> 16 bytes to the host -> allocate -> bin -> sort ->
        # composite: the GPU idles while the host turns around, and the lists take the classic scan + binning chain).
        # The first call of a shape (b, v, G, d_sh, h, w) runs in exact mode; the following ones run under
        # `plan_pair_budget(that call, slack=auto_plan)` with direct bins, verified by the forward itself (check="early":
        # the verdict is final behind the projection kernel and is copied out there; the host waits for it only after the
        # sort and the compositing have been issued -- ONE wait per call as before, but the GPU works through it).  A call
        # whose plan failed is re-run in exact mode on the spot and re-planned: results are ALWAYS those of exact mode
        # (bit-identical in the forward: the sorted lists are unique), never NaN, nothing is raised.  Evaluation calls go
        # through the graph cache above.  (Verifying at the END of the forward instead was measured: 0.540 ms per C2 step
        # against exact mode's 0.470 -- the host then starts issuing the loss and the backward only when the GPU is idle.)
        # `auto_plan_defer = True` (SPF_AUTO_PLAN_DEFER=1), opt-in, drops even that synchronisation for TRAINING calls: the
        # verdict is copied to pinned memory behind the forward and read at the NEXT call (an event long past by then);
        # if the plan had failed, that one step's images and gradients were all NaN -- what the reference's NaN-gradient
        # guard (model_wrapper.py:1117-1151) skips -- and the next call is exact again and re-plans.
        self._auto_plan: Optional[float] = None
        self.auto_plan = _auto_plan_from_env()
        self.auto_plan_defer = os.environ.get("SPF_AUTO_PLAN_DEFER", "0") == "1"
        self._auto_verdict = None        # the one pinned word + event all of them use
        # TRAINING calls (planned, something requires grad): what the GPU runs per call is ~9 kernels of 5 - 130 us; what the
        # host runs to launch them the general way -- validation, a dozen allocations, argument structs, autograd
        # bookkeeping of 21 saved tensors -- approaches that on a slow host.  So, when a call's SHAPES repeat (all nine input
        # shapes, which inputs require grad, image size, plan), the module PREPARES the step once (rasterizer.StaticStep:
        # the chain's state at fixed addresses, argument structs built once) and later calls bind their inputs -- a dozen
        # pointer fields: an encoder's fresh tensors of every step are as good as static leaves -- and are five C-ABI calls.
        # Outputs and gradients are fresh tensors per call -- nothing a caller holds is rewritten, leaves accumulate as ever
        # --; results are bit-identical to the general path.  A forward issued while another of the same shapes waits for
        # its backward gets a step of its own (two per shape, three per decoder); beyond that the general path.
        # `prepare_steps = False` / SPF_PREPARE_STEPS=0: off.
        self.prepare_steps = os.environ.get("SPF_PREPARE_STEPS", "1") != "0"
        self._prepared_steps: dict = {}    # key (shapes, flags, plan) -> [_PreparedStep]
        self._prepare_seen: dict = {}
        self._graphs: dict = {}          # key -> _EvalGraph (insertion-ordered: oldest first)
        self._graph_seen: dict = {}      # key -> None: keys seen once, not yet captured
        self._graph_unused = 0           # captures since the last replay hit

    _EVAL_GRAPH_SLOTS = 4
    _EVAL_GRAPH_MISSES = 8
    _EVAL_GRAPH_BYTES = 2 << 30      # captured evaluation graphs may pin this much device memory in total

    @property
    def auto_plan(self) -> Optional[float]:
        return self._auto_plan

    @auto_plan.setter
    def auto_plan(self, slack) -> None:
        """None / 0: the module stops planning for itself -- and drops the plan it made (the next call is exact mode's,
        not a stale `check="deferred"` plan nobody verifies any more); a slack factor >= 1 otherwise."""
        if slack is not None and slack != 0 and not (float(slack) >= 1.0):
            raise ValueError(f"auto_plan is a slack factor >= 1 (or None / 0 for off), got {slack!r}")
        self._auto_plan = float(slack) if slack else None
        if not self._auto_plan and getattr(self, "_auto_owned", False):
            self._max_pairs, self._auto_owned, self._auto_key, self._auto_pending = None, False, None, None
            self._auto_plans.clear()

    # a decoder is copied (EMA: copy.deepcopy(model)) and pickled (torch.save(model)) like any module -- both go through
    # __getstate__: captured graphs, events and pinned words stay with the original
    _TRANSIENT = ("_graphs", "_graph_seen", "_auto_verdict", "_auto_pending", "_prepared_steps", "_prepare_seen")

    def __getstate__(self):
        state = dict(self.__dict__)
        for k in self._TRANSIENT:
            if k in state:
                state[k] = {} if isinstance(state[k], dict) else None
        state["last_call"], state["_last_call_borrowed"] = CallRecord(), False
        return state

    @property
    def max_pairs(self):
        return self._max_pairs

    @max_pairs.setter
    def max_pairs(self, plan) -> None:           # (the caller's word: the module's own planning stands back until it is None again)
        self._stash_auto()
        self._max_pairs, self._auto_owned, self._auto_key, self._auto_pending = plan, False, None, None

    def _stash_auto(self) -> None:
        """Remember the module's own plan of the current shape (a plan is verified by every call that uses it, so a stale
        one costs one re-run)."""
        if self._auto_owned and self._auto_key is not None and self._max_pairs is not None:
            if len(self._auto_plans) >= 8 and self._auto_key not in self._auto_plans:
                self._auto_plans.pop(next(iter(self._auto_plans)))
            self._auto_plans[self._auto_key] = self._max_pairs

    def _set_auto(self, plan) -> None:
        self._max_pairs, self._auto_owned = plan, True

    def clear_eval_graphs(self) -> None:
        self._graphs.clear()
        self._graph_seen.clear()
        self._graph_unused = 0

    def _own_record(self) -> CallRecord:
        """The record a general-path call writes into: the decoder's own.  After a call on a prepared step or a replayed
        graph `last_call` IS that entry's record (counters, verdict word: what its next call is checked by) -- a general
        call that wrote into it would have the entry verify its next call against another call's buffers."""
        if self._last_call_borrowed:
            self.last_call, self._last_call_borrowed = CallRecord(), False
        return self.last_call

    def clear_prepared_steps(self) -> None:
        self._prepared_steps.clear()
        self._prepare_seen.clear()

    _PREPARED_SLOTS = 4          # prepared steps per decoder (each holds a whole call's state)
    _PREPARED_PER_KEY = 2        # ... of which for the same shapes (forwards waiting for their backward at the same time)

    def _prepare_key(self, tensors, image_shape):
        """None unless this call may run on a prepared step: planned with a list-length class (direct bins), dense
        float32 device tensors, no capture going on.  The key holds SHAPES, not addresses: the inputs are bound per call
        (StaticStep.bind), so an encoder's fresh tensors of every step find the same step.  Training calls (something
        requires grad) and evaluation calls (nothing will be differentiated: forward-only steps, for the calls the graph
        cache cannot serve because their tensors move) have keys of their own."""
        plan = self.max_pairs
        if not (self.prepare_steps and isinstance(plan, PairBudget) and plan.max_tile_list > 0):
            return None
        if torch.cuda.is_current_stream_capturing():
            return None
        grad = torch.is_grad_enabled()
        flags = []
        dev = tensors[0].device
        for t in tensors:
            if t.dtype != torch.float32 or t.device != dev or not t.is_cuda or not t.is_contiguous():
                return None                              # (the general launcher names what is wrong with it)
            flags.append(grad and bool(t.requires_grad))
        if any(flags[1:4]):                             # (intrinsics / near / far are not differentiable inputs)
            return None
        band4 = sh_band4_default() if self.sh_band4 is None else bool(self.sh_band4)
        return (tuple(t.shape for t in tensors), tuple(flags), dev.index, tuple(image_shape),
                int(plan.capacity), int(plan.max_tile_list), band4, self.background_color.data_ptr(),
                self.make_scale_invariant, self.enable_cov_grad, self.enable_sh_grad)

    def _prepare_step(self, key, gaussians, extrinsics, intrinsics, near, far, image_shape):
        """Prepare the static step; None (and prepare_steps off) when that fails."""
        from .rasterizer import StaticStep, _direct_bin_cap, _f32c
        from ._lib import load
        h, w = image_shape
        b, v = extrinsics.shape[:2]
        G = gaussians.means.shape[1]
        high = getattr(gaussians, "harmonics_band4", None)
        n = gaussians.harmonics.shape[-1] + (0 if high is None else high.shape[-1])
        T = load().spf_raster_num_tiles(h, w)
        if not _direct_bin_cap(self.max_pairs, b * v * T, T):
            return None
        # (shapes as render_batch checks them: a mismatch raises here, once, with the usual message)
        _f32c(intrinsics, "intrinsics", (b, v, 3, 3)); _f32c(near, "near", (b, v)); _f32c(far, "far", (b, v))
        _f32c(gaussians.scales, "scales", (b, G, 3)); _f32c(gaussians.rotations, "rotations", (b, G, 4))
        _f32c(gaussians.opacities, "opacities", (b, G))
        _f32c(gaussians.harmonics, "shs", (b, G, 3, 16 if high is not None else n))
        if high is not None:
            _f32c(high, "shs_high", (b, G, 3, 9))
        band4 = sh_band4_default() if self.sh_band4 is None else bool(self.sh_band4)
        trains = any(key[1])
        want = dict(scales_rot=trains and self.enable_cov_grad and (gaussians.scales.requires_grad or gaussians.rotations.requires_grad),
                    shs=trains and self.enable_sh_grad and gaussians.harmonics.requires_grad, colors=False,
                    view=trains and bool(extrinsics.requires_grad), means2D=False)
        # room: the oldest step no forward is waiting on makes way; none such -> this call the general way
        while sum(map(len, self._prepared_steps.values())) >= self._PREPARED_SLOTS:
            victim = next(((k, e) for k, es in self._prepared_steps.items() for e in es if not e.busy()), None)
            if victim is None:
                return None
            self._prepared_steps[victim[0]].remove(victim[1])
            if not self._prepared_steps[victim[0]]:
                del self._prepared_steps[victim[0]]
        try:
            with torch.no_grad(), torch.cuda.device(extrinsics.device):
                step = StaticStep(extrinsics, intrinsics, near, far, gaussians.means, gaussians.scales, gaussians.rotations,
                                  gaussians.opacities, gaussians.harmonics, high, self.background_color, h, w, isqrt(n) - 1,
                                  self.make_scale_invariant, self.max_pairs, band4, want, forward_only=not trains)
                entry = _PreparedStep()
                entry.step, entry.gen, entry.token = step, 0, None
        except Exception as e:                      # noqa: BLE001
            import warnings
            self.prepare_steps = False
            self._prepare_seen.pop(key, None)
            warnings.warn(f"spfsplatv2_amd: preparing a step for this call failed ({type(e).__name__}: {e}); this decoder "
                          "launches its calls the general way from now on")
            return None
        entry.record = CallRecord(counters=step.counters, plan=step.plan_info)
        entry.nbytes = step.nbytes
        self._prepared_steps.setdefault(key, []).append(entry)
        self._prepare_seen.pop(key, None)
        return entry

    def _render_prepared(self, entry, gaussians, extrinsics, intrinsics, near, far, want_extra: bool, trains: bool,
                         image_shape):
        check = self.max_pairs.check
        entry.step.bind(extrinsics, intrinsics, near, far, gaussians.means, gaussians.scales, gaussians.rotations,
                        gaussians.opacities, gaussians.harmonics, getattr(gaussians, "harmonics_band4", None))
        if not trains:
            # an evaluation call on a forward-only step: verified like a replayed graph's (the wait is for the projection
            # kernel only); a plan that did not hold for THESE inputs -> this call in exact mode, on a record of its own
            step = entry.step
            color, depth, alpha, failed = step.forward(check != "deferred")
            if failed:
                self.last_call, self._last_call_borrowed = CallRecord(), False
                with torch.no_grad():
                    color, depth, alpha, radii = self._render_eager(gaussians, extrinsics, intrinsics, near, far,
                                                                    image_shape, None, self.last_call)
                return DecoderOutput(color, depth), alpha, radii
            radii = step.radii.view(alpha.shape[0], alpha.shape[1], -1).clone() if want_extra else None
            self.last_call, self._last_call_borrowed = entry.record, True
            return DecoderOutput(color, depth), (alpha if want_extra else None), radii
        color, depth, alpha, radii = _PreparedRender.apply(
            entry, check, want_extra, extrinsics, gaussians.means, gaussians.scales, gaussians.rotations,
            gaussians.opacities, gaussians.harmonics, getattr(gaussians, "harmonics_band4", None))
        self.last_call, self._last_call_borrowed = entry.record, True
        return DecoderOutput(color, depth), alpha, radii

    def _render_eager(self, gaussians, extrinsics, intrinsics, near, far, image_shape, max_pairs, record):
        color, depth, alpha, radii = render_views(
            extrinsics, intrinsics, near, far, image_shape, self.background_color,
            gaussians.means, gaussians.harmonics, gaussians.opacities, gaussians.rotations, gaussians.scales,
            scale_invariant=self.make_scale_invariant, enable_cov_grad=self.enable_cov_grad,
            enable_sh_grad=self.enable_sh_grad, max_pairs=max_pairs, sh_band4=self.sh_band4, return_radii=True,
            record=record, gaussian_sh_band4=getattr(gaussians, "harmonics_band4", None),
            gaussian_raw=getattr(gaussians, "raw", None))
        depth = depth[:, :, 0]                                   # "(b v) 1 h w -> b v h w"
        if self.make_scale_invariant:
            depth = depth * near[:, :, None, None]               # decoder_splatting_cuda.py:72-76
        return color, depth, alpha, radii

    def _eval_graph_key(self, tensors, image_shape):
        """None unless this call may run from a captured graph: planned, nothing will be differentiated, every tensor a
        dense float32 device tensor (what the kernels take without a copy), no capture already going on."""
        if not (self.eval_graphs and isinstance(self.max_pairs, PairBudget)) or os.environ.get("SPF_EVAL_GRAPHS", "1") == "0":
            return None
        grad = torch.is_grad_enabled()
        for t in tensors:
            if (grad and t.requires_grad) or t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
                return None
        if torch.cuda.is_current_stream_capturing():
            return None
        band4 = sh_band4_default() if self.sh_band4 is None else bool(self.sh_band4)
        # (addresses of all nine tensors; the shapes of three of them fix the others' -- (b, v), G and d_sh --, anything
        #  else would have failed the rasterizer's own shape checks on the call that was captured)
        return (tuple(map(_data_ptr, tensors)), tensors[0].shape, tensors[4].shape, tensors[5].shape, tuple(image_shape),
                self.max_pairs, band4, self.background_color.data_ptr(), self.make_scale_invariant)

    def render(self, gaussians: Gaussians, extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
               image_shape: tuple[int, int]):
        """``forward`` plus the two rasterizer outputs the reference's decoder drops (cuda_splatting.py:128,141-144):
        returns (DecoderOutput, alpha [b,v,1,h,w], radii [b,v,g] int32)."""
        return self._render(gaussians, extrinsics, intrinsics, near, far, image_shape, True)

    def _render(self, gaussians, extrinsics, intrinsics, near, far, image_shape, want_extra: bool):
        fused = getattr(gaussians, "raw", None)
        if fused is not None:
            if not (self.enable_cov_grad and self.enable_sh_grad) and torch.is_grad_enabled() and fused.raw.requires_grad:
                raise RuntimeError("Gaussians.raw (adapter fused into the decoder) chains the backward to ALL raw channels: "
                                   "it needs enable_cov_grad and enable_sh_grad")
            # (same positions as below for what the keys look at: [4] means, [5] stands in for the harmonics)
            tensors = (extrinsics, intrinsics, near, far, gaussians.means, fused.raw, gaussians.opacities)
        else:
            tensors = (extrinsics, intrinsics, near, far, gaussians.means, gaussians.harmonics, gaussians.opacities,
                       gaussians.rotations, gaussians.scales)
        if getattr(gaussians, "harmonics_band4", None) is not None:
            tensors = tensors + (gaussians.harmonics_band4,)
        auto = (self.auto_plan and (self._max_pairs is None or self._auto_owned) and extrinsics.is_cuda
                and not torch.cuda.is_current_stream_capturing())
        if not auto:
            return self._render_planned(tensors, gaussians, extrinsics, intrinsics, near, far, image_shape, want_extra)
        # ---- the module's own planning (see __init__) ----
        from .rasterizer import plan_pair_budget
        run = lambda: self._render_planned(tensors, gaussians, extrinsics, intrinsics, near, far, image_shape, want_extra)
        shape = (tuple(extrinsics.shape[:2]), tuple(gaussians.means.shape), tuple(tensors[5].shape), tuple(image_shape))
        if self._auto_pending is not None:
            # (deferred mode) the last training step's verdict, an event long past: a failed plan is neither used again
            # nor remembered for its shape
            verdict, event = self._auto_pending
            self._auto_pending = None
            event.synchronize()
            if int(verdict[0]) != 0:
                self._set_auto(None)
                self._auto_plans.pop(self._auto_key, None)
        if shape != self._auto_key:
            # (a loop may alternate between a few shapes -- training and validation views, test_step's one-view calls:
            #  every shape keeps its plan)
            self._stash_auto()
            self._auto_key = shape
            self._set_auto(self._auto_plans.get(shape))
        trains = torch.is_grad_enabled() and any(t.requires_grad for t in tensors)
        planned = self._max_pairs is not None
        if planned:      # evaluation: "backward" (the forward / the graph path verifies at once); training: "early" (the
            #              forward verifies behind the projection kernel and raises), or nothing at all when deferred
            check = ("deferred" if self.auto_plan_defer else "early") if trains else "backward"
            self._set_auto(self._max_pairs._replace(check=check))
        try:
            result = run()
        except _SpfError:
            if not planned or (trains and self.auto_plan_defer):
                raise
            self._set_auto(None)                         # this call's plan failed: the same call in exact mode, re-plan below
            result = run()
        if self.last_call.get("counters") is None:
            # this call ran in exact mode (the first of its shape, or one that fell back to it) and left statistics:
            # plan the next ones from them
            self._set_auto(plan_pair_budget(self.last_call, slack=float(self.auto_plan), check="deferred"))
        elif trains and self.auto_plan_defer:
            if self._auto_verdict is None:
                self._auto_verdict = (torch.empty(1, dtype=torch.int32, pin_memory=True), torch.cuda.Event())
            verdict, event = self._auto_verdict
            verdict.copy_(self.last_call["counters"][2:3], non_blocking=True)
            event.record()
            self._auto_pending = self._auto_verdict
        return result

    def _render_planned(self, tensors, gaussians, extrinsics, intrinsics, near, far, image_shape, want_extra: bool):
        if getattr(gaussians, "raw", None) is not None:           # (fused adapter: the general launcher, planned or exact)
            color, depth, alpha, radii = self._render_eager(gaussians, extrinsics, intrinsics, near, far, image_shape,
                                                            self.max_pairs, self._own_record())
            return DecoderOutput(color, depth), alpha, radii
        tkey = self._prepare_key(tensors, image_shape)
        trains = tkey is not None and any(tkey[1])
        if tkey is not None:
            # training calls, and evaluation calls on forward-only steps: measured at the test_step shape, a prepared
            # step's five launches into fresh outputs (0.094 ms per call and synchronisation) beat a replay of the same
            # call's captured graph plus the copy-out of its buffers (0.108) -- and serve tensors that move every call,
            # which a cache keyed by addresses never sees twice.  The graph cache below takes what cannot be prepared
            # (plans without direct bins, `prepare_steps = False`).
            entries = self._prepared_steps.get(tkey)
            entry = None
            if entries:
                entry = next((e for e in entries if not e.busy()), None)
                if entry is None and len(entries) < self._PREPARED_PER_KEY:
                    # every step of these shapes is waiting for its backward (one training step renders the target and
                    # the context views with the same shapes, say): one more
                    entry = self._prepare_step(tkey, gaussians, extrinsics, intrinsics, near, far, image_shape)
            elif tkey in self._prepare_seen:
                entry = self._prepare_step(tkey, gaussians, extrinsics, intrinsics, near, far, image_shape)
            else:
                if len(self._prepare_seen) >= 64:
                    self._prepare_seen.clear()
                self._prepare_seen[tkey] = None            # first sight: run as usual; the second call of the shapes is prepared
            if entry is not None:
                return self._render_prepared(entry, gaussians, extrinsics, intrinsics, near, far, want_extra, trains,
                                             image_shape)
        key = None if trains else self._eval_graph_key(tensors, image_shape)
        if key is None:
            color, depth, alpha, radii = self._render_eager(gaussians, extrinsics, intrinsics, near, far, image_shape,
                                                            self.max_pairs, self._own_record())
            return DecoderOutput(color, depth), alpha, radii
        entry = self._graphs.get(key)
        if entry is None:
            first_sight = key not in self._graph_seen
            if first_sight or self._graph_unused >= self._EVAL_GRAPH_MISSES:
                if first_sight:
                    if len(self._graph_seen) >= 64:
                        self._graph_seen.clear()
                    self._graph_seen[key] = None
                with torch.no_grad():
                    color, depth, alpha, radii = self._render_eager(gaussians, extrinsics, intrinsics, near, far,
                                                                    image_shape, self.max_pairs, self._own_record())
                return DecoderOutput(color, depth), alpha, radii
            entry = self._capture(key, gaussians, extrinsics, intrinsics, near, far, image_shape)
            if entry is None:        # the capture failed (another thread's HIP call, out of memory ...): this call eagerly
                with torch.no_grad():
                    color, depth, alpha, radii = self._render_eager(gaussians, extrinsics, intrinsics, near, far,
                                                                    image_shape, self.max_pairs, self._own_record())
                return DecoderOutput(color, depth), alpha, radii
        else:
            self._graph_unused = 0
        verdict = entry.record.get("verdict_host") if entry.record.get("verdict_mirrored") else None
        if verdict is not None:
            verdict.zero_()                          # (pinned host word the graph's projection kernel stores to on failure)
        entry.graph.replay()
        self.last_call, self._last_call_borrowed = entry.record, True
        flat, alpha, radii = entry.outputs
        # ONE copy-out, queued right behind the replay (before any wait for the verdict): what the caller gets is the
        # caller's (the graph's own buffers are rewritten by its next replay); colour and depth were packed into one flat
        # buffer inside the graph
        color, depth = flat.clone().split_with_sizes(entry.sizes)
        extra = (alpha.clone(), radii.clone()) if want_extra else (None, None)
        if self.max_pairs.check != "deferred":
            from .rasterizer import plan_flags
            if verdict is not None:
                # direct bins: wait for the replay, read the host-mapped word -- no device->host copy of the counters
                torch.cuda.current_stream(extrinsics.device).synchronize()
                failed = verdict.item() != 0
            else:
                failed = plan_flags(entry.record) != 0
            if failed:
                # the plan did not hold for THESE inputs (the graph's outputs are NaN): this call in exact mode instead
                # (on a record of its own: the graph's record must keep the counters its next replay is checked by)
                self.last_call, self._last_call_borrowed = CallRecord(), False
                with torch.no_grad():
                    color, depth, alpha, radii = self._render_eager(gaussians, extrinsics, intrinsics, near, far,
                                                                    image_shape, None, self.last_call)
                return DecoderOutput(color, depth), alpha, radii
        out = DecoderOutput(color.view(entry.color_shape), depth.view(entry.depth_shape))   # both contiguous, as ever
        return out, extra[0], extra[1]

    def _capture(self, key, gaussians, extrinsics, intrinsics, near, far, image_shape) -> "Optional[_EvalGraph]":
        """Capture this evaluation call.  None when the capture failed: the key is forgotten, the cache is switched off
        for this decoder (`eval_graphs = False`: a val / test loop that worked before keeps working, kernel by kernel)
        and the caller renders eagerly."""
        while len(self._graphs) >= self._EVAL_GRAPH_SLOTS or \
                (self._graphs and sum(e.nbytes for e in self._graphs.values()) > self._EVAL_GRAPH_BYTES):
            self._graphs.pop(next(iter(self._graphs)))
        try:
            entry = self._capture_unguarded(key, gaussians, extrinsics, intrinsics, near, far, image_shape)
        except Exception as e:                      # noqa: BLE001
            import warnings
            self._graph_seen.pop(key, None)
            self._graphs.pop(key, None)
            self.eval_graphs = False
            warnings.warn(f"spfsplatv2_amd: capturing the evaluation call in a HIP graph failed ({type(e).__name__}: {e}); "
                          "this decoder launches its evaluation calls kernel by kernel from now on")
            return None
        return entry

    def _capture_unguarded(self, key, gaussians, extrinsics, intrinsics, near, far, image_shape) -> "_EvalGraph":
        record = CallRecord(verdict_host=torch.zeros(1, dtype=torch.int32, pin_memory=True))
        graph = torch.cuda.CUDAGraph()
        dev = extrinsics.device
        before = torch.cuda.memory_allocated(dev)
        # (thread_local: a capture-unsafe HIP call of ANOTHER host thread -- a DataLoader's pin_memory thread, another
        #  rank's thread -- does not invalidate this capture)
        with torch.no_grad(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
            color, depth4, alpha, radii = render_views(
                extrinsics, intrinsics, near, far, image_shape, self.background_color,
                gaussians.means, gaussians.harmonics, gaussians.opacities, gaussians.rotations, gaussians.scales,
                scale_invariant=self.make_scale_invariant, enable_cov_grad=self.enable_cov_grad,
                enable_sh_grad=self.enable_sh_grad, max_pairs=self.max_pairs, sh_band4=self.sh_band4, return_radii=True,
                record=record, gaussian_sh_band4=getattr(gaussians, "harmonics_band4", None))
            if self.make_scale_invariant:
                depth4.mul_(near[:, :, None, None, None])            # decoder_splatting_cuda.py:72-76, in place: the buffer is ours
            depth = depth4[:, :, 0]
            # the rasterizer hands out colour and depth as two views of one allocation (rasterizer._forward_impl): that
            # allocation is the one buffer copied out per call; anything else (another binding) is packed here
            if (depth4.untyped_storage().data_ptr() == color.untyped_storage().data_ptr() and color.is_contiguous()
                    and depth4.is_contiguous() and depth4.data_ptr() == color.data_ptr() + 4 * color.numel()):
                flat = color.new_empty(0).set_(color.untyped_storage(), color.storage_offset(),
                                               (color.numel() + depth4.numel(),))
            else:
                flat = torch.cat((color.reshape(-1), depth.reshape(-1)))
        entry = _EvalGraph(graph, (flat, alpha, radii), record)
        entry.nbytes = max(torch.cuda.memory_allocated(dev) - before, 0)     # what the graph's private pool pins
        entry.sizes = [color.numel(), depth.numel()]
        entry.color_shape, entry.depth_shape = tuple(color.shape), tuple(depth.shape)
        self._graphs[key] = entry
        self._graph_seen.pop(key, None)
        self._graph_unused += 1
        return entry

    def forward(self, gaussians: Gaussians, extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                image_shape: tuple[int, int], depth_mode: DepthRenderingMode | None = None) -> DecoderOutput:
        # depth_mode is accepted and ignored, as in the reference (decoder_splatting_cuda.py:49)
        return self._render(gaussians, extrinsics, intrinsics, near, far, image_shape, False)[0]


DecoderSplattingHIP = DecoderSplattingCUDA

DECODERS = {"splatting_cuda": DecoderSplattingCUDA}
DecoderCfg = DecoderSplattingCUDACfg


def get_decoder(decoder_cfg: DecoderCfg) -> Decoder:
    return DECODERS[decoder_cfg.name](decoder_cfg)
