"""CPU oracle for the differentiable Gaussian-splat rasterizer (TEST INFRASTRUCTURE ONLY).

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``spfsplatv2_amd``) never routes through it.

PARITY UNPINNED.  The arithmetic being restated lives in the third-party package
``diff_gauss_pose`` (``git+https://github.com/slothfulxtx/diff-gaussian-rasterization.git@pose``,
/root/reference/requirements.txt:88 -- a floating branch, no SHA), which is not
vendored in /root/reference, cannot be fetched (no network) and has no golden
vectors in the reference (the reference has no tests at all).  What is restated
here is the *published* 3D-Gaussian-splatting rasterisation algorithm
(Kerbl et al. 2023, "3D Gaussian Splatting for Real-Time Radiance Field
Rendering", Sec. 4-6 + appendix) in the conventions the reference's call site
fixes:

* call surface, argument layout, row-vector (transposed) matrices,
  perspective-only ``projmatrix``, differentiable ``viewmatrix``, 6-tuple result:
  /root/reference/src/model/decoder/cuda_splatting.py:105-138
* clip-space convention of ``projmatrix`` (z in (0,1), w = z_view):
  /root/reference/src/model/decoder/cuda_splatting.py:15-42
* SH layout ``[G, K, 3]``: /root/reference/src/model/decoder/cuda_splatting.py:79

Every numbered convention below is "SURVEY.md Appendix B #n".

Everything is plain PyTorch on CPU, generic in dtype (float32 = the comparison
oracle, float64 = the arbiter used to flag knife-edge pixels), and gradients
come from autograd; the three deliberate deviations from naive autograd follow
the 3DGS family and are marked ``[3DGS-grad]``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
from torch import Tensor

TILE = 16  # B#6: 16x16 pixel tiles decide which Gaussians may touch a pixel.

# Real spherical-harmonics constants of the 3DGS family (B#9).
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
         0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435)
# Band 4 (the reference's default d_sh = 25, config/model/encoder/spfsplatv2.yaml:20).  The published 3DGS kernels stop
# at degree 3; whether the `pose` fork evaluates band 4 is unknowable offline, so it is evaluated only
# when asked (`band4=True`).  The constants continue the same real-SH family: index n(n+1)+m, same signs -- pinned by
# tests/golden/sh_basis_goldens.pt, generated from the reference's own table
# /root/reference/src/misc/sht.py::rsh_cart_4 (which also reproduces bands 0-3 above to 1e-15).
SH_C4 = (2.5033429417967046, -1.7701307697799304, 0.9461746957575601,
         -0.6690465435572892, 0.10578554691520431, -0.6690465435572892,
         0.47308734787878004, -1.7701307697799304, 0.6258357354491761)

NEAR_CULL = 0.2          # B#3
LOWPASS = 0.3            # B#5
ALPHA_MAX = 0.99         # B#10
ALPHA_MIN = 1.0 / 255.0  # B#10
T_MIN = 1e-4             # B#10
FOV_CLAMP = 1.3          # B#4

# Knife-edge windows (what `composite(..., want_fragile=True)` / `rasterize` flag; see tests/util.py::compare):
FRAG_ALPHA_REL = 5e-5    # alpha within this relative distance of 1/255 (round 4: was 2e-4) ...
# ... plus what the float32 pixel centre moves ln(alpha) by.  x_pix = ((ndc + 1) W - 1) / 2 with ndc = hom_x / hom_w:
# the quotient and the product are good to ~3 roundings of ndc, i.e. FRAG_POS_NDC * |x - (W-1)/2| pixels, and the sum
# (ndc + 1) W to one rounding at its own size, FRAG_POS_SUM * (W/2 + 1) pixels.  (Round 3 used 1.2e-7 (|x| + W/2 + 1):
# three times this at the image centre, 1.5 times at its edge -- the flagged set was 100-250 x the pixels that
# actually differ.)
FRAG_POS_NDC = 1.8e-7
FRAG_POS_SUM = 6e-8
FRAG_MAG_ULP = 4e-7      # ... plus the float32 rounding of the exponent's own three terms at THEIR size
FRAG_T_REL = 1e-3        # transmittance within this relative distance of the 1e-4 stop


@dataclass
class Projected:
    """Per-Gaussian screen-space record for one view (differentiable fields first)."""
    xy: Tensor          # [G,2] pixel-space centre                                  (B#2)
    depth: Tensor       # [G]   view-space z                                        (B#10)
    conic: Tensor       # [G,3] inverse 2-D covariance (A, B, C): q = A dx^2 + 2 B dx dy + C dy^2
    opacity: Tensor     # [G]
    rgb: Tensor         # [G,3] SH colour (+0.5, clamped at 0) or colors_precomp    (B#9)
    radii: Tensor       # [G] int32, 0 = culled                                     (B#6)
    rect_min: Tensor    # [G,2] int64 tile rect (x,y), inclusive
    rect_max: Tensor    # [G,2] int64 tile rect (x,y), exclusive
    radius_raw: Tensor  # [G] 3*sqrt(lambda_max) before ceil (for knife-edge flagging)
    rgb_raw: Tensor | None = None   # [G,3] SH colour + 0.5 BEFORE the clamp at 0 (for knife-edge flagging)
    depth_tol: Tensor | None = None  # [G] what float32 arithmetic can move `depth` by (for knife-edge flagging: order)


def quat_to_rotmat(q: Tensor) -> Tensor:
    """B#7: q = (r, x, y, z), NOT renormalised.  Returns R_std [G,3,3]."""
    r, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y),
    ], dim=-1).reshape(*q.shape[:-1], 3, 3)


def covariance3d(scales: Tensor, rotations: Tensor, scale_modifier: float) -> Tensor:
    """B#8: Sigma = R S^2 R^T with S = diag(scale_modifier * s)."""
    R = quat_to_rotmat(rotations)
    s = scales * scale_modifier
    RS = R * s[..., None, :]
    return RS @ RS.transpose(-1, -2)


def sh_basis(deg: int, d: Tensor) -> Tensor:
    """Real SH basis values [G, (deg+1)^2] for unit directions d [G,3] (B#9)."""
    x, y, z = d.unbind(-1)
    out = [torch.full_like(x, SH_C0)]
    if deg > 0:
        out += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        out += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2 * zz - xx - yy),
                SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
    if deg > 2:
        out += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z,
                SH_C3[2] * y * (4 * zz - xx - yy),
                SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
                SH_C3[4] * x * (4 * zz - xx - yy), SH_C3[5] * z * (xx - yy),
                SH_C3[6] * x * (xx - 3 * yy)]
    if deg > 3:
        out += [SH_C4[0] * xy * (xx - yy), SH_C4[1] * yz * (3 * xx - yy), SH_C4[2] * xy * (7 * zz - 1),
                SH_C4[3] * yz * (7 * zz - 3), SH_C4[4] * (zz * (35 * zz - 30) + 3),
                SH_C4[5] * xz * (7 * zz - 3), SH_C4[6] * (xx - yy) * (7 * zz - 1),
                SH_C4[7] * xz * (xx - 3 * yy), SH_C4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy))]
    return torch.stack(out, dim=-1)


def camera_position(viewmatrix: Tensor) -> Tensor:
    """B#9: campos c with c @ R + t = 0, R = V[:3,:3] assumed orthonormal -> c = -t @ R^T."""
    return -(viewmatrix[3, :3] @ viewmatrix[:3, :3].transpose(0, 1))


def project(means3D: Tensor, scales: Tensor, rotations: Tensor, opacities: Tensor,
            shs: Tensor | None, colors_precomp: Tensor | None,
            viewmatrix: Tensor, projmatrix: Tensor,
            tanfovx: float, tanfovy: float, H: int, W: int,
            sh_degree: int, scale_modifier: float = 1.0, band4: bool = False,
            frozen: dict | None = None, capture: dict | None = None) -> Projected:
    """Per-Gaussian preprocess (stage R1).

    `capture` / `frozen` make the three [3DGS-grad] conventions testable: a call with `capture={}` records the
    constants each convention treats as fixed (clamped Jacobian coordinates, colour clamp mask, and -- in
    `composite` -- the alpha clamp offset); a call with `frozen=<that dict>` evaluates the plain, fully differentiable
    function in which those values ARE constants.  The documented meaning of the conventions is then: autograd
    gradient of the normal call == true (finite-difference) gradient of the frozen function
    (tests/test_oracle_fd.py)."""
    dt = means3D.dtype
    G = means3D.shape[0]
    Rv, tv = viewmatrix[:3, :3], viewmatrix[3, :3]
    t = means3D @ Rv + tv                                   # B#1 row-vector convention
    tz = t[:, 2]
    in_front = tz > NEAR_CULL                               # B#3

    hom = torch.cat([t, torch.ones_like(t[:, :1])], dim=-1) @ projmatrix
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * pw[:, None]
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5                # B#2
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    xy = torch.stack([px, py], dim=-1)

    # --- 2-D covariance (EWA splatting), B#4/B#5 ------------------------------------
    fx = W / (2.0 * tanfovx)
    fy = H / (2.0 * tanfovy)
    limx, limy = FOV_CLAMP * tanfovx, FOV_CLAMP * tanfovy
    safe_tz = torch.where(in_front, tz, torch.ones_like(tz))
    txz, tyz = t[:, 0] / safe_tz, t[:, 1] / safe_tz
    # [3DGS-grad] when clamped, the clamped coordinate is a constant (no gradient to t.x,
    # and none to t.z through the clamp product).
    inx, iny = txz.abs() <= limx, tyz.abs() <= limy
    cx, cy = (txz.clamp(-limx, limx) * safe_tz).detach(), (tyz.clamp(-limy, limy) * safe_tz).detach()
    if frozen is not None:
        inx, iny, cx, cy = frozen["jac_clamp"]
    if capture is not None:
        capture["jac_clamp"] = (inx, iny, cx, cy)
    tcx = torch.where(inx, t[:, 0], cx)
    tcy = torch.where(iny, t[:, 1], cy)
    zero = torch.zeros_like(tz)
    J = torch.stack([
        fx / safe_tz, zero, -fx * tcx / (safe_tz * safe_tz),
        zero, fy / safe_tz, -fy * tcy / (safe_tz * safe_tz),
    ], dim=-1).reshape(G, 2, 3)
    Wcv = Rv.transpose(0, 1)                                # column-vector view rotation
    M = J @ Wcv                                             # [G,2,3]
    Sigma = covariance3d(scales, rotations, scale_modifier)
    cov = M @ Sigma @ M.transpose(-1, -2)
    a = cov[:, 0, 0] + LOWPASS
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + LOWPASS
    det = a * c - b * b
    ok = in_front & (det != 0)
    safe_det = torch.where(ok, det, torch.ones_like(det))
    conic = torch.stack([c / safe_det, -b / safe_det, a / safe_det], dim=-1)

    with torch.no_grad():                                   # B#6 integer footprint
        mid = 0.5 * (a + c)
        lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        radius_raw = 3.0 * torch.sqrt(lam)
        radius = torch.ceil(radius_raw)
        gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
        # C-style int() truncation toward zero, then clamp to the grid.
        rminx = torch.trunc((px - radius) / TILE).clamp(0, gx)
        rminy = torch.trunc((py - radius) / TILE).clamp(0, gy)
        rmaxx = torch.trunc((px + radius + TILE - 1) / TILE).clamp(0, gx)
        rmaxy = torch.trunc((py + radius + TILE - 1) / TILE).clamp(0, gy)
        finite = torch.isfinite(px) & torch.isfinite(py) & torch.isfinite(radius)
        ok = ok & finite
        rect_min = torch.stack([rminx, rminy], -1)
        rect_max = torch.stack([rmaxx, rmaxy], -1)
        rect_min = torch.where(ok[:, None], rect_min, torch.zeros_like(rect_min)).long()
        rect_max = torch.where(ok[:, None], rect_max, torch.zeros_like(rect_max)).long()
        area = (rect_max[:, 0] - rect_min[:, 0]) * (rect_max[:, 1] - rect_min[:, 1])
        ok = ok & (area > 0)
        radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int32)
        rect_min = torch.where(ok[:, None], rect_min, torch.zeros_like(rect_min))
        rect_max = torch.where(ok[:, None], rect_max, torch.zeros_like(rect_max))

    # --- colour, B#9 ----------------------------------------------------------------
    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        deg = min(sh_degree, 4 if band4 else 3)     # band 4: carried (stride), evaluated only on request
        K = (deg + 1) ** 2
        v = means3D - camera_position(viewmatrix)
        d = v / v.norm(dim=-1, keepdim=True)
        basis = sh_basis(deg, d)                            # [G,K]
        rgb = torch.einsum("gk,gkc->gc", basis, shs[:, :K, :]) + 0.5
        rgb_raw = rgb.detach()
        if frozen is not None:
            rgb = rgb * frozen["rgb_mask"]
        else:
            rgb = torch.clamp(rgb, min=0.0)                 # [3DGS-grad] gradient masked where clamped
        if capture is not None:
            capture["rgb_mask"] = (rgb_raw >= 0).to(dt)
    with torch.no_grad():
        # Order knife edge (see tile_lists: the sort key is the float32 value of z).  The kernels form z in float64 from
        # the same float32 inputs and round once, as this evaluation does when run in float64: the two float32 keys can
        # only differ where the float64 value sits within rounding noise of the dot product (~1e-15 of its LARGEST term)
        # of a float32 rounding boundary.  `depth_tol` > 0 marks those Gaussians (about one in 10^7); a pixel is flagged
        # when such a Gaussian and a list neighbour of (nearly) the same key both contribute visibly.
        big = (means3D.abs() @ Rv.abs())[:, 2] + tv[2].abs()
        z32f = tz.detach().to(torch.float32)
        z32 = z32f.to(torch.float64)
        ulp = (torch.nextafter(z32f.abs(), torch.full_like(z32f, float("inf"))) - z32f.abs()).to(torch.float64)
        to_boundary = 0.5 * ulp - (tz.to(torch.float64) - z32).abs()       # distance of z to the nearest rounding boundary
        depth_tol = torch.where(to_boundary.abs() <= 1e-13 * big.to(torch.float64) + 1e-300, ulp, torch.zeros_like(ulp)).to(dt)
    return Projected(xy=xy, depth=tz, conic=conic, opacity=opacities.reshape(G),
                     rgb=rgb.to(dt), radii=radii, rect_min=rect_min, rect_max=rect_max,
                     radius_raw=radius_raw, rgb_raw=None if colors_precomp is not None else rgb_raw,
                     depth_tol=depth_tol)


def tile_lists(pr: Projected, H: int, W: int):
    """Yield (tx, ty, ids) with ids sorted front-to-back by (depth bits, Gaussian index) (B#10)."""
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    vis = pr.radii > 0
    # B#10 / SURVEY.md 8a R3: the sort key is `tile << 32 | float_bits(depth)` -- the FLOAT32 bits of view-space z.  Two
    # Gaussians whose depths differ by less than float32 resolves share a key and come in index order; ordering them by
    # their float64 depths instead (rounds 1 - 3) made every such near-tie an "order knife edge" that had to be flagged --
    # a 131,072-entry tile list holds thousands of them per pixel.  The oracle now orders by the float32 value of ITS
    # depth: the same key the contract names, whatever dtype the rest is evaluated in.
    depth = pr.depth.detach().to(torch.float32)
    for ty in range(gy):
        row = vis & (pr.rect_min[:, 1] <= ty) & (pr.rect_max[:, 1] > ty)
        row_ids = torch.nonzero(row).flatten()
        rmin, rmax = pr.rect_min[row_ids, 0], pr.rect_max[row_ids, 0]
        for tx in range(gx):
            ids = row_ids[(rmin <= tx) & (rmax > tx)]
            if ids.numel():
                order = torch.sort(depth[ids], stable=True).indices  # ids ascending -> ties by index
                ids = ids[order]
            yield tx, ty, ids


def composite(pr: Projected, bg: Tensor, H: int, W: int, want_fragile: bool = False,
              frozen: dict | None = None, capture: dict | None = None):
    """Tile-wise front-to-back alpha compositing (stage R6).

    Returns image[3,H,W], depth[1,H,W], alpha[1,H,W] (+ fragile[H,W] bool if asked).
    `frozen` / `capture`: see `project` (here: the offset that turns o*exp(power) into min(0.99, .), per tile).
    """
    dt = pr.xy.dtype
    rows_c = [[None] * ((W + TILE - 1) // TILE) for _ in range((H + TILE - 1) // TILE)]
    rows_d = [[None] * len(rows_c[0]) for _ in rows_c]
    rows_a = [[None] * len(rows_c[0]) for _ in rows_c]
    rows_f = [[None] * len(rows_c[0]) for _ in rows_c]
    for tx, ty, ids in tile_lists(pr, H, W):
        x0, y0 = tx * TILE, ty * TILE
        ys = torch.arange(y0, y0 + TILE, dtype=dt)
        xs = torch.arange(x0, x0 + TILE, dtype=dt)
        pyy, pxx = torch.meshgrid(ys, xs, indexing="ij")
        pixx, pixy = pxx.reshape(-1, 1), pyy.reshape(-1, 1)       # [256,1]
        n_pix = pixx.shape[0]
        if ids.numel() == 0:
            C = bg.to(dt)[None, :].expand(n_pix, 3)
            Dp = torch.zeros(n_pix, dtype=dt)
            Ap = torch.zeros(n_pix, dtype=dt)
            frag = torch.zeros(n_pix, dtype=torch.bool)
        else:
            gxy = pr.xy[ids]
            dx = gxy[None, :, 0] - pixx                            # [256,L]
            dy = gxy[None, :, 1] - pixy
            con = pr.conic[ids]
            power = -0.5 * (con[None, :, 0] * dx * dx + con[None, :, 2] * dy * dy) \
                - con[None, :, 1] * dx * dy
            o = pr.opacity[ids][None, :]
            raw = o * torch.exp(torch.clamp(power, max=0.0))
            # [3DGS-grad] min(0.99, .) is straight-through in the backward pass.
            off = (torch.clamp(raw, max=ALPHA_MAX) - raw).detach()
            if frozen is not None:
                off = frozen["alpha_clamp"][(tx, ty)]
            if capture is not None:
                capture.setdefault("alpha_clamp", {})[(tx, ty)] = off
            alpha = raw + off
            with torch.no_grad():
                valid = (power <= 0) & (alpha >= ALPHA_MIN)
            a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
            incl = torch.cumprod(1.0 - a_eff, dim=1)               # T after each entry
            with torch.no_grad():
                keep = valid & (incl >= T_MIN)      # stop BEFORE the entry that drops T below 1e-4
                # cumprod is monotone, so once an entry is refused all later ones are too
                stopped = torch.cumsum((valid & ~keep).to(torch.int32), dim=1) > 0
                keep = keep & ~stopped
            a_k = torch.where(keep, alpha, torch.zeros_like(alpha))
            incl_k = torch.cumprod(1.0 - a_k, dim=1)
            T_excl = torch.cat([torch.ones_like(incl_k[:, :1]), incl_k[:, :-1]], dim=1)
            w = a_k * T_excl
            T_final = incl_k[:, -1]
            C = w @ pr.rgb[ids] + T_final[:, None] * bg.to(dt)[None, :]
            Dp = w @ pr.depth[ids]
            Ap = 1.0 - T_final                                     # B#11
            if want_fragile:
                with torch.no_grad():
                    rel = FRAG_ALPHA_REL
                    # ... plus what the float32 resolution of the pixel centre itself can move alpha by (FRAG_POS_*:
                    # ~1e-5 px at the image centre, ~3e-5 px at the edge of a 256-px image, 1e-4 px at x = 500); at the
                    # rim of a footprint d(ln alpha)/dx = A dx + B dy is ~5 per pixel
                    ex = FRAG_POS_NDC * (gxy[None, :, 0] - 0.5 * (W - 1)).abs() + FRAG_POS_SUM * (0.5 * W + 1.0)
                    ey = FRAG_POS_NDC * (gxy[None, :, 1] - 0.5 * (H - 1)).abs() + FRAG_POS_SUM * (0.5 * H + 1.0)
                    # ... and what evaluating the exponent itself in float32 costs: its three terms are each rounded
                    # at THEIR size (`mag`), which for a thin splat centred hundreds of pixels away is ~1e2..1e3 while
                    # their sum is ~ -5 (ln alpha moves by up to ~1e-4 there; negligible for ordinary footprints)
                    mag = 0.5 * (con[None, :, 0].abs() * dx * dx + con[None, :, 2].abs() * dy * dy) \
                        + (con[None, :, 1] * dx * dy).abs()
                    win = rel + (con[None, :, 0] * dx + con[None, :, 1] * dy).abs() * ex \
                        + (con[None, :, 2] * dy + con[None, :, 1] * dx).abs() * ey + FRAG_MAG_ULP * mag
                    near_alpha = ((alpha - ALPHA_MIN).abs() < win * ALPHA_MIN) & (power <= 0)
                    # power > 0 (skipped) vs <= 0 can only flip where the three terms cancel to rounding level
                    near_pow = power.abs() <= 1e-5 * mag
                    # a float32 running product over <= a few thousand factors drifts by ~1e-5..1e-4 relative; the
                    # entry that decides is the last one kept OR the first one refused
                    refused = valid & ~keep
                    first_refused = refused & (torch.cumsum(refused.to(torch.int32), dim=1) == 1)
                    near_T = ((incl - T_MIN).abs() < FRAG_T_REL * T_MIN) & (keep | first_refused)
                    frag = (near_alpha | near_pow | near_T).any(dim=1)
                    # not a branch but a resolution limit: the same exponent rounding moves every alpha SMOOTHLY by
                    # alpha * 4e-7 * mag; where that adds up to a visible amount no float32 evaluation of the classic
                    # formula reaches 1e-4 (the oracle's own float32 evaluation is off by 6e-5..9e-5 on the pixels
                    # this flags) -- only under splats centred hundreds of pixels away
                    frag = frag | ((w.detach() * mag).sum(dim=1) * 4e-7 > 3e-5)
                    # the ORDER of two entries is decided by float32 depth bits (B#10): where two of a pixel's
                    # contributors are closer in depth than float32 resolves, either may come first -- flagged where
                    # swapping them would move a colour channel by more than 2e-5 (T a_i a_j |c_i - c_j|)
                    if pr.depth_tol is not None and ids.numel() > 1:
                        zs, zt = pr.depth[ids].detach(), pr.depth_tol[ids]
                        cand = keep | first_refused
                        a_c = torch.where(cand, alpha, torch.zeros_like(alpha))
                        col = pr.rgb[ids].detach()
                        for k in range(1, min(ids.numel(), 4)):
                            # (zt is zero except for a Gaussian whose float32 key is itself on a knife edge: then one ulp)
                            tie = ((zt[k:] + zt[:-k]) > 0) & ((zs[k:] - zs[:-k]).abs() <= (zt[k:] + zt[:-k]))
                            if not bool(tie.any()):
                                continue
                            dcol = (col[k:] - col[:-k]).abs().amax(dim=1)
                            swap = T_excl[:, :-k] * a_c[:, :-k] * a_c[:, k:] * dcol[None, :]
                            # (if one of the two is the entry the 1e-4 stop refuses, the swap decides which of them
                            #  is composited at all)
                            at_stop = first_refused[:, k:] | first_refused[:, :-k]
                            swap = torch.where(at_stop, T_excl[:, :-k] * torch.maximum(a_c[:, :-k], a_c[:, k:]), swap)
                            frag = frag | ((swap > 2e-5) & tie[None, :]).any(dim=1)
                    if capture is not None and "fragile_stats" in capture:      # (diagnostics: which window flags how many pixels)
                        fs = capture["fragile_stats"]
                        base = (near_alpha | near_pow | near_T).any(dim=1)
                        for name, m in (("near_alpha", near_alpha.any(dim=1)), ("near_pow", near_pow.any(dim=1)),
                                        ("near_T", near_T.any(dim=1)),
                                        ("smooth_mag", (w.detach() * mag).sum(dim=1) * 4e-7 > 3e-5),
                                        ("order_or_smooth_only", frag & ~base), ("any", frag)):
                            fs[name] = fs.get(name, 0) + int(m.sum())
            else:
                frag = torch.zeros(n_pix, dtype=torch.bool)
        rows_c[ty][tx] = C.reshape(TILE, TILE, 3)
        rows_d[ty][tx] = Dp.reshape(TILE, TILE)
        rows_a[ty][tx] = Ap.reshape(TILE, TILE)
        rows_f[ty][tx] = frag.reshape(TILE, TILE)
    img = torch.cat([torch.cat(r, dim=1) for r in rows_c], dim=0)[:H, :W]
    dep = torch.cat([torch.cat(r, dim=1) for r in rows_d], dim=0)[:H, :W]
    alp = torch.cat([torch.cat(r, dim=1) for r in rows_a], dim=0)[:H, :W]
    out = (img.permute(2, 0, 1).contiguous(), dep[None], alp[None])
    if want_fragile:
        fr = torch.cat([torch.cat(r, dim=1) for r in rows_f], dim=0)[:H, :W]
        return out + (fr,)
    return out


def rasterize(means3D: Tensor, scales: Tensor, rotations: Tensor, opacities: Tensor,
              shs: Tensor | None, colors_precomp: Tensor | None,
              viewmatrix: Tensor, projmatrix: Tensor, bg: Tensor,
              tanfovx: float, tanfovy: float, H: int, W: int,
              sh_degree: int, scale_modifier: float = 1.0, want_fragile: bool = False, band4: bool = False,
              frozen: dict | None = None, capture: dict | None = None, want_radii_fragile: bool = False):
    """One (scene, view) render: the semantics of one ``GaussianRasterizer(settings)(...)`` call
    (/root/reference/src/model/decoder/cuda_splatting.py:124-138).

    Returns (image[3,H,W], depth[1,H,W], alpha[1,H,W], radii[G] int32[, fragile[H,W]]).
    """
    pr = project(means3D, scales, rotations, opacities, shs, colors_precomp, viewmatrix,
                 projmatrix, tanfovx, tanfovy, H, W, sh_degree, scale_modifier, band4=band4,
                 frozen=frozen, capture=capture)
    out = composite(pr, bg, H, W, want_fragile=want_fragile, frozen=frozen, capture=capture)
    if want_fragile:
        # Tile membership decided by a rounding knife-edge (footprint radius within 1e-4 of an integer, or a rect
        # bound within ~1e-4 px of a tile border): only the tiles whose membership would actually change are
        # tainted, i.e. the difference between the largest and the smallest plausible rect.
        with torch.no_grad():
            fr = out[3].clone()
            gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
            px, py = pr.xy[:, 0].double(), pr.xy[:, 1].double()
            raw = pr.radius_raw.double()
            eps = 1e-4 + 2e-6 * px.abs().clamp(max=1e6)     # float32 pixel centres are good to ~1e-6 relative

            def rect(pxx, pyy, rad):
                x0 = torch.trunc((pxx - rad) / TILE).clamp(0, gx); y0 = torch.trunc((pyy - rad) / TILE).clamp(0, gy)
                x1 = torch.trunc((pxx + rad + TILE - 1) / TILE).clamp(0, gx)
                y1 = torch.trunc((pyy + rad + TILE - 1) / TILE).clamp(0, gy)
                return x0, y0, x1, y1

            r_lo, r_hi = torch.ceil(raw * (1 - 1e-5) - 1e-4), torch.ceil(raw * (1 + 1e-5) + 1e-4)
            # union box: smallest mins / largest maxes; intersection box: the opposite
            ux0, uy0, _, _ = rect(px - eps, py - eps, r_hi)
            _, _, ux1, uy1 = rect(px + eps, py + eps, r_hi)
            ix0, iy0, _, _ = rect(px + eps, py + eps, r_lo)
            _, _, ix1, iy1 = rect(px - eps, py - eps, r_lo)
            vis = (pr.radii > 0) | ((ux1 > ux0) & (uy1 > uy0) & (pr.depth.detach() > NEAR_CULL))
            amb = vis & ((ux0 != ix0) | (uy0 != iy0) | (ux1 != ix1) | (uy1 != iy1))
            # SH colour within 1e-6 of the clamp at 0 (B#9): whether the channel -- and the gradient of its 3K
            # coefficients -- is switched off is decided by the last bit; every pixel the Gaussian can reach is tainted
            if pr.rgb_raw is not None:
                clampy = (pr.radii > 0) & (pr.rgb_raw.double().abs() < 1e-6).any(dim=-1)
                for g in torch.nonzero(clampy).flatten().tolist():
                    rr = float(r_hi[g]) + 2.0
                    y0p, y1p = max(0, int(py[g] - rr)), min(H, int(py[g] + rr) + 2)
                    x0p, x1p = max(0, int(px[g] - rr)), min(W, int(px[g] + rr) + 2)
                    if y1p > y0p and x1p > x0p:
                        fr[y0p:y1p, x0p:x1p] = True
            for g in torch.nonzero(amb).flatten().tolist():
                a0, b0, a1, b1 = int(ux0[g]), int(uy0[g]), int(ux1[g]), int(uy1[g])
                c0, d0, c1, d1 = int(ix0[g]), int(iy0[g]), int(ix1[g]), int(iy1[g])
                box = torch.zeros(gy, gx, dtype=torch.bool)
                box[b0:b1, a0:a1] = True
                if c1 > c0 and d1 > d0:
                    box[d0:d1, c0:c1] = False
                # ... and inside those tiles only the pixels the Gaussian would CONTRIBUTE to if the tile were on its
                # list: power <= 0 and alpha >= 1/255 (with the alpha window's margin).  (Round 3 tainted its whole
                # padded 3-sigma box there; beyond 3 sigma alpha = 0.011 x opacity, below 1/255 unless opacity > 0.35.)
                tiles_px = box.repeat_interleave(TILE, 0).repeat_interleave(TILE, 1)[:H, :W]
                if not bool(tiles_px.any()):
                    continue
                yy, xx = torch.nonzero(tiles_px, as_tuple=True)
                cg = pr.conic[g].detach().double()
                ddx, ddy = px[g] - xx.double(), py[g] - yy.double()
                pw_g = -0.5 * (cg[0] * ddx * ddx + cg[2] * ddy * ddy) - cg[1] * ddx * ddy
                al_g = torch.clamp(pr.opacity[g].detach().double() * torch.exp(pw_g.clamp(max=0.0)), max=ALPHA_MAX)
                hit = (pw_g <= 1e-9) & (al_g >= ALPHA_MIN * (1.0 - 1e-3))
                fr[yy[hit], xx[hit]] = True
        res = (out[0], out[1], out[2], pr.radii, fr)
    else:
        res = (out[0], out[1], out[2], pr.radii)
    return res + (radii_fragile(pr, H, W),) if want_radii_fragile else res


def radii_fragile(pr: Projected, H: int, W: int) -> Tensor:
    """[G] bool: Gaussians whose integer `radii` entry (B#6) is decided by a rounding knife-edge -- 3*sqrt(lambda)
    within ~1e-5 relative of an integer (the ceil), a tile rect whose area flips between zero and non-zero, or a
    depth at the near cull.  Everywhere else the product's radii must equal the oracle's exactly."""
    with torch.no_grad():
        gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
        px, py = pr.xy[:, 0].double(), pr.xy[:, 1].double()
        raw = pr.radius_raw.double()
        tz = pr.depth.detach().double()
        eps = 1e-4 + 2e-6 * px.abs().clamp(max=1e6)
        r_lo, r_hi = torch.ceil(raw * (1 - 1e-5) - 1e-4), torch.ceil(raw * (1 + 1e-5) + 1e-4)

        def area(pxl, pxh, pyl, pyh, rad):
            x0 = torch.trunc((pxl - rad) / TILE).clamp(0, gx); y0 = torch.trunc((pyl - rad) / TILE).clamp(0, gy)
            x1 = torch.trunc((pxh + rad + TILE - 1) / TILE).clamp(0, gx)
            y1 = torch.trunc((pyh + rad + TILE - 1) / TILE).clamp(0, gy)
            return (x1 - x0).clamp(min=0) * (y1 - y0).clamp(min=0)

        big = area(px - eps, px + eps, py - eps, py + eps, r_hi) > 0
        small = area(px + eps, px - eps, py + eps, py - eps, r_lo) > 0
        near = (tz - NEAR_CULL).abs() < 1e-5
        odd = ~(torch.isfinite(px) & torch.isfinite(py) & torch.isfinite(raw))
        return (((r_lo != r_hi) | (big != small)) & (tz > NEAR_CULL - 1e-5)) | near | odd


def num_pairs(pr: Projected) -> int:
    """D = number of (Gaussian, tile) pairs of one render (SURVEY.md section 8 byte model)."""
    a = (pr.rect_max - pr.rect_min)
    return int((a[:, 0] * a[:, 1]).sum())
