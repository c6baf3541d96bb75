"""Seeded random sweep of shapes / scales / cameras: HIP path vs float64 oracle (same gates as test_gpu_raster)."""
import pytest
import torch

from spfsplatv2_amd import synthetic as syn
from tests import util

pytestmark = pytest.mark.gpu


def _random_case(seed: int):
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    S, V = ri(1, 3), ri(1, 3)
    K = [1, 4, 9, 16, 25][ri(0, 4)]
    G = ri(1, 2500)
    hw = (ri(5, 90), ri(5, 120))
    s_mult = [1.0, 4.0, 15.0, 60.0, 250.0][ri(0, 4)]
    bg = tuple(float(x) for x in torch.rand(3, generator=g))
    si = bool(ri(0, 1))
    batch = syn.make_batch("TEST", S, V, seed=seed, s_mult=s_mult, G=G, K=K, image_hw=hw)
    # move some cameras so that part of the scene is behind / very close to the camera, vary near per view
    batch.extrinsics[..., 2, 3] += (torch.rand(S, V, generator=g) - 0.3) * 3.0
    batch.near = batch.near * (0.5 + torch.rand(S, V, generator=g) * 2.0)
    batch.opacities = (batch.opacities * (0.2 + 1.0 * torch.rand(1, generator=g))).clamp(max=0.999)
    return batch, bg, si, dict(S=S, V=V, K=K, G=G, hw=hw, s_mult=s_mult, si=si)


@pytest.mark.parametrize("seed", list(range(100, 116)) + [204, 248, 275])   # + an empty render, huge splats, an SH clamp edge
def test_random_configurations(hip_lib, seed):
    batch, bg, si, desc = _random_case(seed)
    ref = util.run_oracle(batch, torch.float64, background=bg, scale_invariant=si, mask_fragile=True)
    prod = util.run_product(batch, background=bg, scale_invariant=si, pixel_mask=ref["pixel_mask"])
    # (tiny images under splats hundreds of pixels wide: up to a tenth of the pixels can sit on a knife edge)
    rep = util.compare(prod, ref, max_fragile_frac=0.10)
    assert not rep["fails"], (desc, rep)


@pytest.mark.parametrize("seed", [2135, 2195, 2389])
def test_wide_seeds_that_exposed_the_float32_determinant(hip_lib, seed):
    """Three draws of the fuzz campaign's `wide` family (footprints x 60 ... x 250: thin splats hundreds of pixels
    long) on which det = a*c - b*b of the 2-D covariance lost its leading digits in float32 and up to 350 pixels were
    off by 3e-4 ... 2.4e-3.  The projection kernel now forms det without that cancellation
    (project.hip::stable_det) and the view-space position in float64; the oracle has NO flag for this (round 2 had
    masked 63 % / 9 % / 80 % of these images).  Seeds 2195 and 2389 must pass every gate with the ordinary masks.
    Seed 2135 holds a zero-thickness needle (two scales exactly 0) 4,300 px long and 0.55 px thin whose centre lies
    1,170 px outside the image: one unit in the last place of its float32 pixel centre moves the exponent of the
    pixels it crosses by ~5e-4, and its dL/dmean is a sum of +- terms over ~100 pixels that cancels to ~1 % -- no
    float32 evaluation resolves that (the oracle's own float32 evaluation is 4.5e-4 off in colour and 15 % off in that
    gradient).  There a gate may be missed only where the oracle's float32 evaluation misses it too
    (`util.float32_resolvable`), and by no more than 3x its error; the kernel is at 1.0e-4 / 15 %."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_campaign", util.ROOT / "tools" / "fuzz_campaign.py")
    fc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fc)
    batch, bg, si, band4, _planned, desc = fc.random_case(seed, wide=True)
    ref = util.run_oracle(batch, torch.float64, background=bg, scale_invariant=si, mask_fragile=True, band4=band4)
    prod = util.run_product(batch, background=bg, scale_invariant=si, pixel_mask=ref["pixel_mask"], band4=band4)
    rep = util.compare(prod, ref, max_fragile_frac=0.10)
    if seed != 2135:
        assert not rep["fails"], (desc, rep)
        return
    if rep["fails"]:
        r32 = util.float32_resolvable(batch, ref, background=bg, scale_invariant=si, band4=band4)
        # the needle's own rotation gradient (it reaches dL/dq only through the one column of R its single non-zero
        # scale keeps: the large n n^T part of dL/dcov cancels there) is the one gate the float32 oracle meets and the
        # kernel does not (1.2e-2): it must be confined to the Gaussians with two zero scales
        needles = (batch.scales == 0).sum(dim=-1) >= 2                       # [S,G]
        keep = (~needles)[..., None].to(torch.float64)
        rot_err = ((prod["grads"]["rotations"].double() - ref["grads"]["rotations"]) * keep).abs().max()
        assert float(rot_err / ref["grads"]["rotations"].abs().max()) < 1e-3, rot_err
        allowed = set(r32["fails"]) | {"g_rotations"}
        assert set(rep["fails"]) <= allowed, (desc, rep, r32)
        for f in rep["fails"]:
            assert rep[f] <= (3.0 * r32[f] if f in r32["fails"] else 2e-2), (f, rep[f], r32[f])
