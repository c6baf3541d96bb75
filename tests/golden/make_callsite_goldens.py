"""Generate call-site golden fixtures by importing the REFERENCE's own Python glue.

Runs ONLY in the build container (needs /root/reference); the GPU box never sees the reference.
Output: tests/golden/callsite_*.pt -- inputs plus exactly what the reference hands to its rasterizer
(`GaussianRasterizationSettings` fields and `GaussianRasterizer.__call__` kwargs) for each (scene, view),
captured with a recording stand-in for the un-vendored `diff_gauss_pose` package, and the
post-processing the reference applies to the rasterizer's depth output.

    python tests/golden/make_callsite_goldens.py
"""
import importlib.util
import sys
import types
from pathlib import Path
from typing import NamedTuple

import torch

REF = Path("/root/reference/src")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(OUT.parents[1]))

# ---- minimal environment so that the reference modules import --------------------------------
jt = types.ModuleType("jaxtyping")


class _Ann:
    def __class_getitem__(cls, item):
        return cls


for _n in ("Float", "Int64", "Bool", "UInt8", "Shaped", "Int"):
    setattr(jt, _n, type(_n, (_Ann,), {}))
sys.modules["jaxtyping"] = jt

calls = []
dg = types.ModuleType("diff_gauss_pose")


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    projmatrix: torch.Tensor
    sh_degree: int
    prefiltered: bool
    debug: bool
    enable_cov_grad: bool
    enable_sh_grad: bool


class GaussianRasterizer:
    def __init__(self, settings):
        self.s = settings

    def __call__(self, **kw):
        calls.append((self.s, kw))
        h, w, g = self.s.image_height, self.s.image_width, kw["means3D"].shape[0]
        # a recognisable depth so that the reference's post-processing (x near) can be captured
        depth = torch.full((1, h, w), 2.0) + len(calls)
        return (torch.zeros(3, h, w), depth, torch.zeros(3, h, w), torch.zeros(1, h, w),
                torch.zeros(g, dtype=torch.int32), None)


dg.GaussianRasterizationSettings = GaussianRasterizationSettings
dg.GaussianRasterizer = GaussianRasterizer
sys.modules["diff_gauss_pose"] = dg


def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


for p in ("src", "src.geometry", "src.model", "src.model.decoder"):
    _pkg(p)
_pkg("src.dataset").DatasetCfg = object


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, REF / rel)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


_load("src.geometry.projection", "geometry/projection.py")
types_mod = _load("src.model.types", "model/types.py")
_load("src.model.decoder.decoder", "model/decoder/decoder.py")
cs = _load("src.model.decoder.cuda_splatting", "model/decoder/cuda_splatting.py")
dsc = _load("src.model.decoder.decoder_splatting_cuda", "model/decoder/decoder_splatting_cuda.py")


def _snap(settings, kw):
    d = {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in settings._asdict().items()}
    d["tanfovx"] = float(d["tanfovx"])
    d["tanfovy"] = float(d["tanfovy"])
    a = {k: (None if v is None else v.detach().clone()) for k, v in kw.items()}
    a.pop("means2D")
    return {"settings": d, "kwargs": a,
            "projmatrix_is_contiguous": bool(settings.projmatrix.is_contiguous()),
            "viewmatrix_requires_grad": bool(kw["viewmatrix"].requires_grad)}


def main():
    from spfsplatv2_amd import synthetic as syn

    # 1) decoder path: b=2 scenes x v=3 views, d_sh = 4 (degree 1), different near per view
    for tag, K, scale_inv in (("decoder_k4_si", 4, True), ("decoder_k25_nosi", 25, False)):
        calls.clear()
        b = syn.make_batch("C1", 2, 3, seed=11, K=K, s_mult=20.0)
        near = b.near * torch.tensor([[1.0, 2.0, 0.5], [1.5, 1.0, 3.0]])
        far = b.far * torch.tensor([[1.0, 2.0, 0.5], [1.5, 1.0, 3.0]])
        ext = b.extrinsics.clone().requires_grad_(True)
        cfg = dsc.DecoderSplattingCUDACfg(name="splatting_cuda", background_color=[0.1, 0.2, 0.3],
                                          make_scale_invariant=scale_inv, enable_cov_grad=True, enable_sh_grad=True)
        dec = dsc.DecoderSplattingCUDA(cfg)
        g = types_mod.Gaussians(b.means, b.covariances, b.rotations, b.scales, b.harmonics, b.opacities)
        out = dec.forward(g, ext, b.intrinsics, near, far, b.image_shape)
        torch.save({
            "inputs": {"means": b.means, "covariances": b.covariances, "rotations": b.rotations, "scales": b.scales,
                       "harmonics": b.harmonics, "opacities": b.opacities, "extrinsics": b.extrinsics,
                       "intrinsics": b.intrinsics, "near": near, "far": far, "image_shape": b.image_shape,
                       "background_color": cfg.background_color, "make_scale_invariant": scale_inv},
            "calls": [_snap(s, kw) for s, kw in calls],
            "decoder_depth": out.depth.detach().clone(),   # fake depth (2 + call#) after the reference's x near
            "decoder_color_shape": tuple(out.color.shape),
        }, OUT / f"callsite_{tag}.pt")
        print(tag, len(calls), "calls")

    # 2) render_cuda with use_sh=False (colors_precomp) and orthographic
    calls.clear()
    b = syn.make_batch("C1", 3, 1, seed=12, K=1)
    img, dep = cs.render_cuda(b.extrinsics[:, 0], b.intrinsics[:, 0], b.near[:, 0], b.far[:, 0], b.image_shape,
                              torch.rand(3, 3, generator=torch.Generator().manual_seed(0)), b.means, b.covariances,
                              b.harmonics, b.opacities, b.rotations, b.scales, scale_invariant=True, use_sh=False)
    snap_persp = [_snap(s, kw) for s, kw in calls]
    calls.clear()
    # the reference's orthographic path only works for batch 1 (`move_back[2, 3] = -distance_to_near`,
    # cuda_splatting.py:184, needs a one-element tensor): capture one call per item
    width = torch.tensor([2.0, 3.0, 4.0])
    height = torch.tensor([2.0, 1.5, 4.0])
    for i in range(3):
        cs.render_cuda_orthographic(b.extrinsics[i:i + 1, 0], width[i:i + 1], height[i:i + 1], b.near[i:i + 1, 0],
                                    b.far[i:i + 1, 0], (48, 64), torch.zeros(1, 3), b.means[i:i + 1],
                                    b.covariances[i:i + 1], b.harmonics[i:i + 1], b.opacities[i:i + 1],
                                    b.rotations[i:i + 1], b.scales[i:i + 1], fov_degrees=0.1, use_sh=True)
    snap_ortho = [_snap(s, kw) for s, kw in calls]
    torch.save({
        "inputs": {"means": b.means, "covariances": b.covariances, "rotations": b.rotations, "scales": b.scales,
                   "harmonics": b.harmonics, "opacities": b.opacities, "extrinsics": b.extrinsics[:, 0],
                   "intrinsics": b.intrinsics[:, 0], "near": b.near[:, 0], "far": b.far[:, 0],
                   "image_shape": b.image_shape, "bg": torch.rand(3, 3, generator=torch.Generator().manual_seed(0)),
                   "ortho_width": width, "ortho_height": height, "ortho_image_shape": (48, 64)},
        "calls_precomp": snap_persp, "calls_ortho": snap_ortho,
    }, OUT / "callsite_render_cuda.pt")
    print("render_cuda", len(snap_persp), "ortho", len(snap_ortho))

    # 3) get_fov / get_projection_matrix tables
    gen = torch.Generator().manual_seed(5)
    K = syn.intrinsics(6)
    K[:, 0, 0] += torch.rand(6, generator=gen) * 0.5
    K[:, 1, 1] += torch.rand(6, generator=gen) * 0.5
    K[:, 0, 2] += (torch.rand(6, generator=gen) - 0.5) * 0.1
    fov = sys.modules["src.geometry.projection"].get_fov(K)
    near = torch.rand(6, generator=gen) + 0.1
    far = near + torch.rand(6, generator=gen) * 100
    proj = cs.get_projection_matrix(near, far, fov[:, 0], fov[:, 1])
    torch.save({"intrinsics": K, "fov": fov, "near": near, "far": far, "proj": proj}, OUT / "camera_tables.pt")


if __name__ == "__main__":
    main()
