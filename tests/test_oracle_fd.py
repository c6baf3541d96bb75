"""Narrowing the unpinned-oracle gap (SURVEY.md 7.2 "verify with finite differences in float64 on
the oracle"): oracle/splat_ref.py's gradients -- which every HIP gradient is gated against -- are checked against
central finite differences of its own forward in float64, for all six differentiable inputs of the rasterizer call
(means3D, scales, rotations, opacities, shs, viewmatrix: cuda_splatting.py:128-138).

Two regimes:
* away from the three deliberate [3DGS-grad] conventions (no alpha at the 0.99 clamp, no Jacobian clamp, no colour
  clamped at 0) the oracle is an ordinary differentiable function: autograd == finite differences;
* with all three conventions ACTIVE, autograd must equal the finite differences of the function in which exactly the
  documented quantities are constants (splat_ref's `capture` / `frozen` hooks): the clamped alpha's offset to
  o*exp(power), the clamped Jacobian coordinates, the colour clamp mask.  That is the explicit statement of the
  conventions; naive finite differences of the plain forward must NOT match there (otherwise the case tests nothing).
"""
import pytest
import torch

from oracle import glue_ref, splat_ref
from spfsplatv2_amd import synthetic as syn

NAMES = ("means3D", "scales", "rotations", "opacities", "shs", "viewmatrix")
H = W = 64


def _case(seed, zoom=1.0, s_mult=30.0, opacity_scale=0.9, dc_shift=0.8, G=256):
    b = syn.make_batch("C1", 1, 1, seed=seed, s_mult=s_mult, K=4, G=G)
    intr = b.intrinsics.clone()
    intr[..., 0, 0] *= zoom
    intr[..., 1, 1] *= zoom
    harm = b.harmonics.clone()
    harm[..., 0] = harm[..., 0] * 0.3 + dc_shift
    a = glue_ref.callsite_args(b.extrinsics[:, 0], intr[:, 0], b.near[:, 0], b.far[:, 0], (H, W),
                               torch.tensor([[0.2, 0.4, 0.6]]), b.means, harm, b.opacities * opacity_scale,
                               b.rotations, b.scales)[0]
    x = {k: a[k].double() for k in NAMES}
    gen = torch.Generator().manual_seed(seed + 1)
    target = torch.rand(3, H, W, generator=gen, dtype=torch.float64)
    wd, wa = torch.rand(H, W, generator=gen, dtype=torch.float64), torch.rand(H, W, generator=gen, dtype=torch.float64)

    def loss(x, **kw):
        img, dep, alp, _ = splat_ref.rasterize(x["means3D"], x["scales"], x["rotations"], x["opacities"], x["shs"],
                                               None, x["viewmatrix"], a["projmatrix"].double(), a["bg"].double(),
                                               a["tanfovx"], a["tanfovy"], H, W, a["sh_degree"], **kw)
        return ((img - target) ** 2).mean() + 0.01 * (dep[0] * wd).mean() + 0.1 * (alp[0] * wa).mean()
    return x, loss


def _autograd(x, loss, **kw):
    leaves = {k: v.clone().requires_grad_(True) for k, v in x.items()}
    loss(leaves, **kw).backward()
    return {k: v.grad for k, v in leaves.items()}


def _directional_fd(x, loss, name, direction, eps, **kw):
    with torch.no_grad():
        xp, xm = dict(x), dict(x)
        xp[name] = x[name] + eps * direction
        xm[name] = x[name] - eps * direction
        return float((loss(xp, **kw) - loss(xm, **kw)) / (2 * eps))


def _check(x, loss, grads, seed, tol, **kw):
    gen = torch.Generator().manual_seed(seed)
    worst = {}
    for name in NAMES:
        g = grads[name] / grads[name].norm().clamp(min=1e-300)
        for k in range(3):
            # the gradient direction itself, then two random directions that keep a component along it (so that the
            # directional derivative stays well above the rounding noise of a float64 central difference)
            d = torch.randn(x[name].shape, generator=gen, dtype=torch.float64)
            d = g if k == 0 else 0.6 * g + 0.8 * d / d.norm()
            d = d / d.norm()
            # (a viewmatrix step moves EVERY Gaussian: keep it small enough that no alpha crosses 1/255 on the way)
            eps = 1e-6 if name == "viewmatrix" else 1e-6 * max(1.0, float(x[name].abs().max()))
            fd = _directional_fd(x, loss, name, d, eps, **kw)
            an = float((grads[name] * d).sum())
            err = abs(fd - an) / max(abs(an), abs(fd), 1e-12)
            worst[name] = max(worst.get(name, 0.0), err)
    bad = {k: v for k, v in worst.items() if v > tol}
    return worst, bad


def test_autograd_matches_finite_differences_in_the_smooth_regime():
    x, loss = _case(seed=11, zoom=0.6)           # wide field of view: every centre inside 1.3 x tanfov
    cap = {}
    loss(x, capture=cap)
    inx, iny, _, _ = cap["jac_clamp"]
    assert bool(inx.all()) and bool(iny.all())                               # Jacobian clamp inactive
    assert bool((cap["rgb_mask"] == 1).all())                                # no colour clamped at 0
    assert all(float(o.abs().max()) == 0.0 for o in cap["alpha_clamp"].values())   # no alpha at 0.99
    grads = _autograd(x, loss)
    assert all(float(g.abs().max()) > 0 for g in grads.values())
    worst, bad = _check(x, loss, grads, seed=5, tol=2e-6)
    assert not bad, worst


def test_the_three_gradient_conventions_are_straight_through_forms():
    # zoomed camera (many centres beyond 1.3 x tanfov, large splats still reach the image), opaque Gaussians
    # (alpha hits 0.99), DC colours around zero (clamped channels)
    x, loss = _case(seed=12, zoom=2.2, s_mult=120.0, opacity_scale=1.3, dc_shift=-1.7)
    x["opacities"] = x["opacities"].clamp(max=0.9995)
    cap = {}
    loss(x, capture=cap)
    inx, iny, _, _ = cap["jac_clamp"]
    n_jac = int((~inx | ~iny).sum())
    n_rgb = int((cap["rgb_mask"] == 0).sum())
    n_alpha = sum(int((o != 0).sum()) for o in cap["alpha_clamp"].values())
    assert n_jac > 5 and n_rgb > 5 and n_alpha > 5, (n_jac, n_rgb, n_alpha)  # every convention is exercised
    grads = _autograd(x, loss)
    # (a) autograd == finite differences of the function with the documented quantities held constant
    worst, bad = _check(x, loss, grads, seed=6, tol=2e-6, frozen=cap)
    assert not bad, worst
    # (b) and it is NOT the naive derivative of the plain forward: the case would otherwise prove nothing
    naive, _ = _check(x, loss, grads, seed=6, tol=2e-6)
    assert max(naive.values()) > 1e-3, naive


@pytest.mark.parametrize("band4", [False, True])
def test_sh_direction_gradient_incl_band4(band4):
    """Higher SH bands make the colour depend on the view direction (mean - campos): FD over means3D / viewmatrix /
    shs with 25 coefficients, band 4 on and off."""
    b = syn.make_batch("TEST", 1, 1, seed=21, s_mult=25.0, K=25, G=120, image_hw=(32, 32))
    harm = b.harmonics.clone()
    harm[..., 0] = harm[..., 0] * 0.2 + 1.0
    harm[..., 1:] *= 4.0
    a = glue_ref.callsite_args(b.extrinsics[:, 0], b.intrinsics[:, 0], b.near[:, 0], b.far[:, 0], (32, 32),
                               torch.zeros(1, 3), b.means, harm, b.opacities * 0.9, b.rotations, b.scales)[0]
    assert a["sh_degree"] == 4 and a["shs"].shape == (120, 25, 3)
    x = {k: a[k].double() for k in NAMES}
    tgt = torch.rand(3, 32, 32, generator=torch.Generator().manual_seed(2), dtype=torch.float64)

    def loss(x, **kw):
        img = splat_ref.rasterize(x["means3D"], x["scales"], x["rotations"], x["opacities"], x["shs"], None,
                                  x["viewmatrix"], a["projmatrix"].double(), a["bg"].double(), a["tanfovx"],
                                  a["tanfovy"], 32, 32, 4, band4=band4, **kw)[0]
        return ((img - tgt) ** 2).mean()
    cap = {}
    loss(x, capture=cap)
    assert bool((cap["rgb_mask"] == 1).all())
    grads = _autograd(x, loss)
    assert (float(grads["shs"][:, 16:].abs().max()) > 0) == band4
    global H, W
    worst, bad = _check(x, loss, grads, seed=8, tol=5e-6)
    assert not bad, worst
