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


def test_split_adapter_writes_the_same_numbers_into_two_planes(hip_lib):
    """UnifiedGaussianAdapter(split_harmonics=True): harmonics [.,3,16] + harmonics_band4 [.,3,9] are the reference
    layout's [.,3,25] bit for bit (pinned by the goldens above), gradients likewise; a band-4 plane that receives NO
    gradient (degree-3 decoder) gives exact zeros in the raw channels of band 4; and the raw channels are read IN PLACE
    from a strided view of an 83-channel head output (encoder_spfsplatv2.py:261-268) -- no contiguous copy."""
    from spfsplatv2_amd import adapter
    gen = torch.Generator("cuda").manual_seed(3)
    cfg = adapter.GaussianAdapterCfg(0.5, 15.0, 4)
    dense, split = adapter.UnifiedGaussianAdapter(cfg).cuda(), adapter.UnifiedGaussianAdapter(cfg, split_harmonics=True).cuda()
    head = torch.randn(2, 3, 1000, 83, device="cuda", generator=gen)         # [b, v, r, 1 + 82]: density first
    head[..., 1:4] *= 8.0                                                      # (some scales reach the 0.3 clamp)
    means, opac = torch.randn(2, 3, 1000, 1, 1, 3, device="cuda", generator=gen), torch.rand(2, 3, 1000, 1, 1, device="cuda", generator=gen)
    view = head.clone().requires_grad_(True)
    raw_v = view[..., 1:].reshape(2, 3, 1000, 1, 1, 82)                       # "b v r srf c -> b v r srf () c" of the slice
    assert raw_v.data_ptr() == view.data_ptr() + 4
    contiguous = head[..., 1:].contiguous().reshape(2, 3, 1000, 1, 1, 82).requires_grad_(True)
    a = dense(means, opac, contiguous, with_covariances=False)
    b = split(means, opac, raw_v, with_covariances=False)
    assert a.harmonics_band4 is None and tuple(b.harmonics.shape[-2:]) == (3, 16) and tuple(b.harmonics_band4.shape[-2:]) == (3, 9)
    assert torch.equal(b.scales, a.scales) and torch.equal(b.rotations, a.rotations)
    assert torch.equal(torch.cat((b.harmonics, b.harmonics_band4), dim=-1), a.harmonics)
    w = [torch.randn_like(t) for t in (a.scales, a.rotations, a.harmonics)]
    ((a.scales * w[0]).sum() + (a.rotations * w[1]).sum() + (a.harmonics * w[2]).sum()).backward()
    ((b.scales * w[0]).sum() + (b.rotations * w[1]).sum() + (b.harmonics * w[2][..., :16]).sum()
     + (b.harmonics_band4 * w[2][..., 16:]).sum()).backward()
    assert float(view.grad[..., 0].abs().max()) == 0.0                        # the density channel is not the adapter's
    assert torch.equal(view.grad[..., 1:].reshape(contiguous.shape), contiguous.grad)
    # a decoder that evaluates to degree 3: no gradient for the band-4 plane -> zeros for its raw channels, the rest as before
    view.grad = None
    b = split(means, opac, raw_v, with_covariances=False)
    ((b.scales * w[0]).sum() + (b.rotations * w[1]).sum() + (b.harmonics * w[2][..., :16]).sum()).backward()
    got = view.grad[..., 1:].reshape(-1, 82)
    want = contiguous.grad.reshape(-1, 82).clone()
    sh = want[:, 7:].view(-1, 3, 25)
    sh[..., 16:] = 0.0
    assert torch.equal(got, want)


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
