"""Fused HIP Gaussian adapter vs outputs and gradients of the reference's UnifiedGaussianAdapter.forward
(tests/golden/make_adapter_goldens.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("deg", [0, 2, 4])
def test_adapter_matches_reference(hip_lib, golden_dir, deg):
    from spfsplatv2_amd import adapter
    g = torch.load(golden_dir / "adapter_goldens.pt")[f"deg{deg}"]
    ad = adapter.UnifiedGaussianAdapter(adapter.GaussianAdapterCfg(0.5, 15.0, deg)).cuda()
    assert torch.allclose(ad.sh_mask.cpu(), g["sh_mask"])
    raw = g["raw"].cuda().requires_grad_(True)
    out = ad(g["means"].cuda(), g["opacities"].cuda(), raw)          # covariances by default, like the reference
    for name in ("scales", "rotations", "harmonics", "covariances"):
        got, want = getattr(out, name).detach().cpu(), g[name]
        assert got.shape == want.shape, name
        assert float((got - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max())), name
    w = [t.cuda() for t in g["weights"]]
    ((out.scales * w[0]).sum() + (out.rotations * w[1]).sum() + (out.harmonics * w[2]).sum()).backward()
    err = float((raw.grad.cpu() - g["raw_grad"]).abs().max())
    assert err <= 1e-5 * max(1.0, float(g["raw_grad"].abs().max())), err
    lean = ad(g["means"].cuda(), g["opacities"].cuda(), g["raw"].cuda(), with_covariances=False)
    assert lean.covariances.shape == g["covariances"].shape and lean.covariances.stride()[-1] == 0   # not materialised
    assert bool(torch.isnan(lean.covariances).all())                                                  # and loud if used


def test_adapter_feeds_decoder(hip_lib):
    """adapter -> decoder end to end on the device: gradients reach the raw network channels."""
    import spfsplatv2_amd as spf
    from spfsplatv2_amd import adapter, synthetic as syn
    b = syn.make_batch("TEST", 2, 2, seed=51, G=500, K=4, image_hw=(48, 48)).to("cuda")
    ad = adapter.UnifiedGaussianAdapter(adapter.GaussianAdapterCfg(0.5, 15.0, 1)).cuda()
    raw = torch.randn(2, 500, ad.d_in, device="cuda", generator=torch.Generator("cuda").manual_seed(1)).requires_grad_(True)
    g = ad(b.means, b.opacities, raw, with_covariances=False)
    dec = spf.get_decoder(spf.DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True, True, True)).cuda()
    out = dec(spf.Gaussians(g.means, g.covariances, g.rotations, g.scales * 5.0, g.harmonics, g.opacities),
              b.extrinsics, b.intrinsics, b.near, b.far, b.image_shape)
    ((out.color - b.target) ** 2).mean().backward()
    assert bool(torch.isfinite(raw.grad).all()) and float(raw.grad.abs().max()) > 0
