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


# Knife-edge budget per seed: TWICE the fraction of pixels the float64 oracle flags on that seed (measured on the CPU,
# round 5; the mask is the oracle's alone), at least 0.2 % (one pixel of a 500-pixel image).  Rounds 2 - 4 allowed every
# in-suite seed 10 % -- the campaigns' "inconclusive" threshold -- while the fixed cases were capped at 0.2 - 0.5 %.
FRAGILE_MEASURED = {100: 3.0e-4, 102: 5.6e-4, 104: 1.2e-4, 105: 8.2e-5, 106: 5.7e-3, 111: 1.9e-5, 112: 2.9e-2,
                    114: 2.2e-4, 115: 5.3e-3, 248: 1.62e-2, 275: 1.21e-2,
                    "wide2135": 2.02e-2, "wide2195": 1.03e-2, "wide2389": 3.75e-2, "wide170586": 2.5e-2,
                    "wide260130": 2.43e-2, "wide460960": 7.0e-3}      # (seeds not listed: nothing flagged)


def fragile_cap(key) -> float:
    return max(2.0 * FRAGILE_MEASURED.get(key, 0.0), 0.002)


@pytest.mark.parametrize("seed", list(range(100, 116)) + [204, 248, 275])   # + an empty render, huge splats, an SH clamp edge
def test_random_configurations(hip_lib, seed):
    batch, bg, si, desc = _random_case(seed)
    ref = util.run_oracle(batch, torch.float64, background=bg, scale_invariant=si, mask_fragile=True)
    prod = util.run_product(batch, background=bg, scale_invariant=si, pixel_mask=ref["pixel_mask"])
    rep = util.compare(prod, ref, max_fragile_frac=fragile_cap(seed))
    assert not rep["fails"], (desc, rep)


def _wide_case(seed):
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_campaign", util.ROOT / "tools" / "fuzz_campaign.py")
    fc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fc)
    batch, bg, si, band4, _planned, desc = fc.random_case(seed, wide=True)
    ref = util.run_oracle(batch, torch.float64, background=bg, scale_invariant=si, mask_fragile=True, band4=band4)
    prod = util.run_product(batch, background=bg, scale_invariant=si, pixel_mask=ref["pixel_mask"], band4=band4)
    rep = util.compare(prod, ref, max_fragile_frac=fragile_cap(f"wide{seed}"))
    return batch, bg, si, band4, desc, ref, rep


@pytest.mark.parametrize("seed", [2135, 2195, 2389])
def test_wide_seeds_that_exposed_the_float32_determinant(hip_lib, seed):
    """Three draws of the fuzz campaign's `wide` family (footprints x 60 ... x 250: thin splats hundreds of pixels
    long) on which det = a*c - b*b of the 2-D covariance lost its leading digits in float32 and up to 350 pixels were
    off by 3e-4 ... 2.4e-3.  The projection kernel now forms det without that cancellation
    (project.hip::stable_det) and the view-space position in float64; the oracle has NO flag for this (round 2 had
    masked 63 % / 9 % / 80 % of these images).  Every gate, ordinary masks, NO exemption -- also for seed 2135 (a
    zero-thickness needle 4,300 px long, centred 1,170 px outside the image), which round 3 let through on a
    hand-made allowance: it meets every gate since round 4 (colour 1.2e-6, gradients 2.1e-4 of scale; report case
    `wide_seed_2135` in profiles/r04_parity_reports.jsonl)."""
    *_, desc, _ref, rep = _wide_case(seed)
    if seed == 2135:
        from tests.test_gpu_raster import _report
        _report("wide_seed_2135", rep)
    assert not rep["fails"], (desc, rep)


@pytest.mark.parametrize("seed", [170586, 260130])
def test_wide_seeds_at_the_edge_of_the_per_element_gate(hip_lib, seed):
    """The two disagreements of round 4's fuzz campaigns (profiles/r04_fuzz_campaign_b.json, _e.json; 34,909 cases):
    world scale 250, ONE entry of dL/dmeans that is >= 1 % of the tensor's largest off by 2.9e-2 / 1.6e-2 of itself
    (gate `gel_means` 1e-2 -- an addition of this suite to north_star's 1e-3-of-scale gate, which both cases meet:
    7.2e-4 / 1.8e-4), everything else inside its gate.  They lived only in a JSON; here they are named cases that
    report the product's error NEXT TO the error of the same restatement evaluated in float32 on the CPU
    (`util.float32_resolvable`: same masks, same gates), and every gate except that one is enforced.  The float32
    oracle's own `gel_means` on these inputs depends on the summation order (6.7e-3 / 1.0e-3 with 8 threads; above the
    gate on the 256-core host that classified them "unresolvable" in round 4), so it is reported, not used as a bound;
    the product's entry is held at 1.5 x what the campaigns measured."""
    batch, bg, si, band4, desc, ref, rep = _wide_case(seed)
    f32 = util.float32_resolvable(batch, ref, background=bg, scale_invariant=si, band4=band4)
    from tests.test_gpu_raster import _report
    _report(f"wide_seed_{seed}", {**rep, **{"oracle_f32_" + k: v for k, v in f32.items() if k.startswith(("g_", "gel_", "rgb_max"))}})
    assert set(rep["fails"]) <= {"gel_means"}, (desc, rep)
    assert rep["gel_means"] <= 1.5 * {170586: 2.9e-2, 260130: 1.6e-2}[seed], (desc, rep, f32)
    assert rep["g_means"] <= 1e-3


def test_wide_seed_with_one_pixel_on_the_alpha_threshold_of_a_far_needle(hip_lib):
    """The one disagreement of round 6's fuzz campaigns (profiles/r06_fuzz_campaign_c.json; 30,096 cases on the final
    kernels): ONE pixel of a 5 x 249 image, under a needle (sigma 0.74 x 530 px) centred 1,545 px outside it whose alpha
    there is within 2e-4 of 1/255.  The exponent's three terms are 145, -269 and 133 (they sum to 8.6); the product's conic
    is good to 3 ulp of each entry, which moves ln(alpha) by up to 1.3e-4 -- more than the oracle's knife-edge window allots
    at that pixel (1.6e-4 in all, of which 1.1e-4 for the ROUNDING of those terms, none for the conic's own 3 ulp).  So the
    pixel is not flagged, the product drops (or keeps) that one contributor and differs by exactly one threshold-level
    layer: alpha by ~1/255, colour by ~1e-3.  A float32 evaluation of the oracle's own formulas happens to land on the
    float64 side here (`oracle_f32` in the campaign file), so the case counts as a disagreement, stated as such -- and is
    held here to what it is: one pixel, one layer, every other gate met."""
    batch, bg, si, band4, desc, ref, rep = _wide_case(460960)
    from tests.test_gpu_raster import _report
    _report("wide_seed_460960", rep)
    assert set(rep["fails"]) <= {"rgb_max", "alpha_max"}, (desc, rep)
    pixels = batch.extrinsics.shape[0] * batch.extrinsics.shape[1] * batch.image_shape[0] * batch.image_shape[1]
    assert rep["bad_frac_all"] * pixels <= 2.5, (desc, rep)                       # (the unmasked count: this pixel and at most one flagged one)
    assert 0.8 / 255 <= rep["alpha_max"] <= 1.01 / 255 and rep["rgb_max"] <= 1.5e-3, (desc, rep)
    assert max(rep[k] for k in rep if k.startswith("g_")) <= 1e-3 and rep["radii_mismatch"] == 0, (desc, rep)
