"""Seeded synthetic scenes with the value distributions of the reference's producers.

Used by tests, bench.py and smoke(); CPU tensors come out, the caller moves them to the device.

* cameras: normalised intrinsics fx = fy = 0.889, cx = cy = 0.5 (RE10K 640x360 cropped to a square,
  /root/reference/src/dataset/shims/crop_shim.py:44-75), context baseline normalised to 1
  (/root/reference/src/dataset/dataset_re10k.py:178-187), near 0.1 / far 100 (dataset_re10k.py:65-66).
* Gaussians are "pixel aligned" like the encoder's output
  (/root/reference/src/model/encoder/encoder_spfsplatv2.py:248-268): one per pixel of each context grid,
  depth log-uniform in [1, 20]; scales 0.001*softplus(N(0,1)) clamped at 0.3, unit-normalised random
  quaternions (/root/reference/src/model/encoder/common/gaussian_adapter.py:132-136), opacity
  sigmoid(N(0,1)), SH DC ~ N(0,1) and higher bands x 0.1*0.25^deg (gaussian_adapter.py:47-48).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
from torch import Tensor

FX = 0.889


def intrinsics(n: int) -> Tensor:
    K = torch.eye(3).repeat(n, 1, 1)
    K[:, 0, 0] = FX
    K[:, 1, 1] = FX
    K[:, 0, 2] = 0.5
    K[:, 1, 2] = 0.5
    return K


def _rot(axis: Tensor, angle: Tensor) -> Tensor:
    axis = axis / axis.norm()
    x, y, z = axis.tolist()
    Kx = torch.tensor([[0, -z, y], [z, 0, -x], [-y, x, 0]], dtype=torch.float32)
    return torch.eye(3) + math.sin(angle) * Kx + (1 - math.cos(angle)) * (Kx @ Kx)


def small_pose(gen: torch.Generator, baseline: float, max_angle_deg: float = 8.0) -> Tensor:
    """Camera-to-world pose: rotation by <= max_angle about a random axis, translation of length `baseline`
    mostly sideways (x/y), a little along z."""
    axis = torch.randn(3, generator=gen)
    ang = float(torch.rand(1, generator=gen)) * math.radians(max_angle_deg)
    t = torch.randn(3, generator=gen) * torch.tensor([1.0, 0.4, 0.25])
    t = t / t.norm() * baseline
    c2w = torch.eye(4)
    c2w[:3, :3] = _rot(axis, ang)
    c2w[:3, 3] = t
    return c2w


@dataclass
class Scene:
    means: Tensor        # [G,3]
    scales: Tensor       # [G,3]
    rotations: Tensor    # [G,4]
    opacities: Tensor    # [G]
    harmonics: Tensor    # [G,3,K]   (reference layout, encoder side)
    covariances: Tensor  # [G,3,3]   zeros (dead input, kept for signature parity)


def pixel_aligned_scene(gen: torch.Generator, G: int, grid_hw: tuple[int, int], n_grids: int, K: int,
                        s_mult: float = 1.0) -> tuple[Scene, Tensor]:
    """Returns (scene, context c2w [n_grids,4,4])."""
    h, w = grid_hw
    ctx = [torch.eye(4)] + [small_pose(gen, 1.0) for _ in range(n_grids - 1)]
    ctx = torch.stack(ctx)
    pts = []
    v, u = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")
    for g in range(n_grids):
        z = torch.exp(torch.rand(h * w, generator=gen) * math.log(20.0))
        x = (u.reshape(-1) - 0.5) / FX * z
        y = (v.reshape(-1) - 0.5) / FX * z
        cam = torch.stack([x, y, z], dim=-1)
        pts.append(cam @ ctx[g, :3, :3].T + ctx[g, :3, 3])
    means = torch.cat(pts)[:G]
    assert means.shape[0] == G, (means.shape, G)
    scales = (0.001 * torch.nn.functional.softplus(torch.randn(G, 3, generator=gen))).clamp(max=0.3) * s_mult
    q = torch.randn(G, 4, generator=gen)
    q = q / q.norm(dim=-1, keepdim=True)
    opac = torch.sigmoid(torch.randn(G, generator=gen))
    sh = torch.randn(G, 3, K, generator=gen)
    deg = 0
    for k in range(K):
        if k >= (deg + 1) ** 2:
            deg += 1
        if deg > 0:
            sh[:, :, k] *= 0.1 * 0.25 ** deg
    return Scene(means, scales, q, opac, sh, torch.zeros(G, 3, 3)), ctx


def target_poses(gen: torch.Generator, n: int) -> Tensor:
    """Target views between / around the context cameras (baseline <= 1)."""
    return torch.stack([small_pose(gen, float(torch.rand(1, generator=gen)) * 0.8 + 0.1, 6.0) for _ in range(n)])


@dataclass
class Batch:
    """b scenes x v target views, everything the decoder call needs."""
    means: Tensor        # [b,G,3]
    scales: Tensor       # [b,G,3]
    rotations: Tensor    # [b,G,4]
    opacities: Tensor    # [b,G]
    harmonics: Tensor    # [b,G,3,K]
    covariances: Tensor  # [b,G,3,3]
    extrinsics: Tensor   # [b,v,4,4] c2w
    intrinsics: Tensor   # [b,v,3,3]
    near: Tensor         # [b,v]
    far: Tensor          # [b,v]
    image_shape: tuple[int, int]
    target: Tensor       # [b,v,3,h,w] U(0,1): MSE target (loss_mse.py:48-51)

    def to(self, device) -> "Batch":
        kw = {k: (v.to(device) if isinstance(v, Tensor) else v) for k, v in self.__dict__.items()}
        return Batch(**kw)


CONFIGS = {
    # name: (G, image hw, grid hw, n_grids, K)
    "C1": (256, (64, 64), (16, 16), 1, 1),
    "C2": (65536, (256, 256), (256, 256), 1, 1),
    "C3": (320000, (256, 256), (256, 256), 5, 1),
    "C5": (500000, (512, 512), (512, 512), 2, 16),
    # what the shipped 2-view model hands its decoder: one Gaussian per context pixel of two 256x256 views, 25 SH
    # coefficients per channel (/root/reference/src/model/encoder/encoder_spfsplatv2.py:240,296-321;
    # config/model/encoder/spfsplatv2.yaml:20)
    "REF2V": (131072, (256, 256), (256, 256), 2, 25),
    # what the shipped 10-view model hands its decoder: ten 256x256 context grids = 655,360 Gaussians per scene, 25 SH
    # coefficients, rendered to ONE target view per scene, three scenes per step
    # (/root/reference/config/experiment/spfsplatv2/re10k_10view.yaml:36-37,48: num_context_views 10,
    # num_target_views 1, batch_size 3) -- few tiles (768 per step), lists of thousands of entries
    "REF10V": (655360, (256, 256), (256, 256), 10, 25),
    # small parity-test scenes (oracle finishes in seconds): up to 2 x 64 x 64 Gaussians
    "TEST": (4096, (64, 64), (64, 64), 2, 1),
    # many Gaussians per tile (exercises the long-list sort classes): up to 2 x 256 x 256 Gaussians
    "TESTBIG": (70000, (32, 32), (256, 256), 2, 1),
}


def make_batch(config: str, n_scenes: int, n_views: int, seed: int, s_mult: float = 1.0,
               G: int | None = None, image_hw: tuple[int, int] | None = None, K: int | None = None) -> Batch:
    G0, hw, grid, n_grids, K0 = CONFIGS[config]
    G = G or G0
    hw = image_hw or hw
    K = K or K0
    if G > grid[0] * grid[1] * n_grids:
        raise ValueError("G exceeds the pixel-aligned grid capacity")
    gen = torch.Generator().manual_seed(seed)
    scenes, ext = [], []
    for _ in range(n_scenes):
        sc, _ctx = pixel_aligned_scene(gen, G, grid, n_grids, K, s_mult)
        scenes.append(sc)
        ext.append(target_poses(gen, n_views))
    st = lambda f: torch.stack([getattr(s, f) for s in scenes])
    b, v = n_scenes, n_views
    return Batch(
        means=st("means"), scales=st("scales"), rotations=st("rotations"), opacities=st("opacities"),
        harmonics=st("harmonics"), covariances=st("covariances"),
        extrinsics=torch.stack(ext), intrinsics=intrinsics(b * v).reshape(b, v, 3, 3),
        near=torch.full((b, v), 0.1), far=torch.full((b, v), 100.0), image_shape=hw,
        target=torch.rand(b, v, 3, *hw, generator=gen))


def pairs_per_render_estimate(G: int) -> int:
    return 2 * G
