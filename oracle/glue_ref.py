"""CPU restatement of the reference's decoder glue (TEST INFRASTRUCTURE ONLY -- see splat_ref.py header).

Restates, on CPU tensors and independently of the product package:
  * get_fov                      /root/reference/src/geometry/projection.py:269-283
  * get_projection_matrix        /root/reference/src/model/decoder/cuda_splatting.py:15-42
  * render_cuda's argument prep  /root/reference/src/model/decoder/cuda_splatting.py:63-120
  * DecoderSplattingCUDA.forward /root/reference/src/model/decoder/decoder_splatting_cuda.py:41-78
Pinned by tests/golden/callsite_*.pt and camera_tables.pt, which were captured from the reference's
own Python (tests/golden/make_callsite_goldens.py).
"""
from __future__ import annotations

from math import isqrt

import torch
from torch import Tensor

from . import splat_ref


def fov_from_intrinsics(K: Tensor) -> Tensor:
    """projection.py:269-283: angle between normalised K^-1 rays through opposite edge midpoints."""
    Kinv = torch.linalg.inv(K)

    def ray(v):
        d = Kinv @ torch.tensor(v, dtype=K.dtype)
        return d / d.norm(dim=-1, keepdim=True)

    fx = (ray([0.0, 0.5, 1.0]) * ray([1.0, 0.5, 1.0])).sum(-1).acos()
    fy = (ray([0.5, 0.0, 1.0]) * ray([0.5, 1.0, 1.0])).sum(-1).acos()
    return torch.stack([fx, fy], dim=-1)


def projection_matrix(near: Tensor, far: Tensor, fov_x: Tensor, fov_y: Tensor) -> Tensor:
    """cuda_splatting.py:15-42."""
    tx, ty = (0.5 * fov_x).tan(), (0.5 * fov_y).tan()
    top, right = ty * near, tx * near
    bottom, left = -top, -right
    P = torch.zeros(near.shape[0], 4, 4, dtype=torch.float32)
    P[:, 0, 0] = 2 * near / (right - left)
    P[:, 1, 1] = 2 * near / (top - bottom)
    P[:, 0, 2] = (right + left) / (right - left)
    P[:, 1, 2] = (top + bottom) / (top - bottom)
    P[:, 3, 2] = 1
    P[:, 2, 2] = far / (far - near)
    P[:, 2, 3] = -(far * near) / (far - near)
    return P


def callsite_args(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, image_shape,
                  background: Tensor, means: Tensor, harmonics: Tensor, opacities: Tensor, rotations: Tensor,
                  scales: Tensor, scale_invariant: bool = True, use_sh: bool = True) -> list[dict]:
    """What the reference hands to its rasterizer for every item of a FLAT batch (cuda_splatting.py:63-138).

    All arguments carry a leading batch dim B (one Gaussian set per item).  Returns B dicts with
    the settings fields and call kwargs.
    """
    if scale_invariant:                                      # cuda_splatting.py:66-74
        scale = 1 / near
        extrinsics = extrinsics.clone()
        extrinsics[..., :3, 3] = extrinsics[..., :3, 3] * scale[:, None]
        means = means * scale[:, None, None]
        scales = scales * scale[:, None, None]
        near = near * scale
        far = far * scale
    n = harmonics.shape[-1]
    degree = isqrt(n) - 1                                    # cuda_splatting.py:77-78
    shs = harmonics.permute(0, 1, 3, 2).contiguous()         # b g xyz n -> b g n xyz
    fov = fov_from_intrinsics(intrinsics)
    fov_x, fov_y = fov.unbind(-1)
    tan_x, tan_y = (0.5 * fov_x).tan(), (0.5 * fov_y).tan()
    proj = projection_matrix(near, far, fov_x, fov_y).transpose(-1, -2)
    view = torch.linalg.inv(extrinsics).transpose(-1, -2)
    h, w = image_shape
    out = []
    for i in range(extrinsics.shape[0]):
        out.append(dict(
            image_height=h, image_width=w, tanfovx=float(tan_x[i]), tanfovy=float(tan_y[i]), bg=background[i],
            scale_modifier=1.0, projmatrix=proj[i], sh_degree=degree,
            means3D=means[i], shs=shs[i] if use_sh else None,
            colors_precomp=None if use_sh else shs[i, :, 0, :], opacities=opacities[i, ..., None],
            scales=scales[i], rotations=rotations[i], viewmatrix=view[i]))
    return out


def orthographic_callsite_args(extrinsics, width, height, near, far, image_shape, background, means, harmonics,
                               opacities, rotations, scales, fov_degrees=0.1, use_sh=True) -> list[dict]:
    """cuda_splatting.py:146-255 (argument preparation), batch-1 semantics applied per item."""
    b = extrinsics.shape[0]
    n = harmonics.shape[-1]
    degree = isqrt(n) - 1
    shs = harmonics.permute(0, 1, 3, 2).contiguous()
    fov_x = torch.tensor(fov_degrees).deg2rad()
    tan_x = (0.5 * fov_x).tan()
    dist = (0.5 * width) / tan_x
    tan_y = 0.5 * height / dist
    fov_y = (2 * tan_y).atan()
    near = near + dist
    far = far + dist
    # cuda_splatting.py:183-201 builds inverse(extrinsics @ move_back), move_back = translate(0, 0, -dist).  That
    # equals translate(0, 0, +dist) @ inverse(extrinsics) -- the same matrix without a float32 inverse of a pose whose
    # translation is ~10^3 units (that inverse is only good to ~1e-7 of the translation: pixel centres would wobble by
    # ~1e-3 px between one inverse routine and the next).  Pinned against the reference's own call-site vectors
    # (tests/golden/callsite_render_cuda.pt, `calls_ortho`) like the rest of this file.
    w2c = torch.linalg.inv(extrinsics).clone()
    w2c[:, 2, 3] = w2c[:, 2, 3] + dist
    proj = projection_matrix(near, far, fov_x.expand(b), fov_y).transpose(-1, -2)
    view = w2c.transpose(-1, -2)
    h, w = image_shape
    return [dict(image_height=h, image_width=w, tanfovx=float(tan_x), tanfovy=float(tan_y[i]), bg=background[i],
                 scale_modifier=1.0, projmatrix=proj[i], sh_degree=degree, means3D=means[i],
                 shs=shs[i] if use_sh else None, colors_precomp=None if use_sh else shs[i, :, 0, :],
                 opacities=opacities[i, ..., None], scales=scales[i], rotations=rotations[i], viewmatrix=view[i])
            for i in range(b)]


def decoder_forward(means, harmonics, opacities, rotations, scales, extrinsics, intrinsics, near, far, image_shape,
                    background_color, make_scale_invariant=True, dtype=torch.float32, want_fragile=False,
                    band4=False, want_radii_fragile=False):
    """DecoderSplattingCUDA.forward (decoder_splatting_cuda.py:41-78) on the CPU oracle.

    [b,g,...] Gaussians, [b,v,...] cameras -> color [b,v,3,h,w], depth [b,v,h,w] (already x near),
    alpha [b,v,1,h,w], radii [b,v,g] (+ fragile [b,v,h,w]) (+ radii_fragile [b,v,g]).  Differentiable w.r.t. its
    float inputs.  `band4`: evaluate SH band 4 when d_sh = 25 (see splat_ref.SH_C4).
    """
    b, v = extrinsics.shape[:2]
    rep = lambda t: t[:, None].expand(b, v, *t.shape[1:]).reshape(b * v, *t.shape[1:])   # the `repeat`s
    bg = torch.as_tensor(background_color, dtype=torch.float32)
    args = callsite_args(extrinsics.reshape(b * v, 4, 4), intrinsics.reshape(b * v, 3, 3), near.reshape(-1),
                         far.reshape(-1), image_shape, bg[None].expand(b * v, 3), rep(means), rep(harmonics),
                         rep(opacities), rep(rotations), rep(scales), scale_invariant=make_scale_invariant)
    cols, deps, alps, rads, frs, rfr = [], [], [], [], [], []
    for a in args:
        c = lambda t: None if t is None else t.to(dtype)
        out = splat_ref.rasterize(c(a["means3D"]), c(a["scales"]), c(a["rotations"]), c(a["opacities"]),
                                  c(a["shs"]), c(a["colors_precomp"]), c(a["viewmatrix"]), c(a["projmatrix"]),
                                  c(a["bg"]), a["tanfovx"], a["tanfovy"], a["image_height"], a["image_width"],
                                  a["sh_degree"], a["scale_modifier"], want_fragile=want_fragile, band4=band4,
                                  want_radii_fragile=want_radii_fragile)
        cols.append(out[0]); deps.append(out[1]); alps.append(out[2]); rads.append(out[3])
        if want_fragile:
            frs.append(out[4])
        if want_radii_fragile:
            rfr.append(out[-1])
    h, w = image_shape
    color = torch.stack(cols).reshape(b, v, 3, h, w)
    depth = torch.stack(deps).reshape(b, v, h, w)
    if make_scale_invariant:
        depth = depth * near.to(dtype)[:, :, None, None]                     # decoder_splatting_cuda.py:72-76
    alpha = torch.stack(alps).reshape(b, v, 1, h, w)
    radii = torch.stack(rads).reshape(b, v, -1)
    res = (color, depth, alpha, radii)
    if want_fragile:
        res += (torch.stack(frs).reshape(b, v, h, w),)
    if want_radii_fragile:
        res += (torch.stack(rfr).reshape(b, v, -1),)
    return res
