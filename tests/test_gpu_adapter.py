"""Fused HIP Gaussian adapter vs outputs and gradients of the reference's UnifiedGaussianAdapter.forward
(tests/golden/make_adapter_goldens.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import util  # noqa: E402


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


@pytest.mark.parametrize("deg,views,strided", [(4, 2, True), (1, 3, False), (0, 1, False), (4, 9, False)],
                         ids=["d_sh25_view_of_head", "d_sh4", "d_sh1", "d_sh25_nine_views"])
def test_adapter_fused_into_the_decoder_is_the_two_pass_path(hip_lib, deg, views, strided):
    """`UnifiedGaussianAdapter(fuse_into_decoder=True)` + `DecoderSplattingCUDA` (SpfDims.sh_layout 3: the projection
    kernels apply the adapter's activations as they read a raw row and chain the backward through them) against the
    adapter's own kernels followed by the decoder on their outputs: the same expressions in the same order, so images,
    depth, alpha and EVERY gradient -- to the raw network channels, the means, the opacities, the poses -- agree to
    float32 rounding (another instantiation of the kernels: the compiler may contract a multiply-add differently; 2e-6 /
    2e-5 of scale here, north_star asks for 1e-4 / 1e-3) and the radii exactly, with the raw rows read in place from a
    strided view of an 83-channel head output (encoder_spfsplatv2.py:261-268); nine views: the gradient of the harmonics
    leaves through the unstaged path."""
    import spfsplatv2_amd as spf
    from spfsplatv2_amd import adapter, synthetic as syn
    K = (deg + 1) ** 2
    b = syn.make_batch("TEST", 2, views, seed=61 + deg, s_mult=8.0, G=900, K=K, image_hw=(64, 48)).to("cuda")
    gen = torch.Generator("cuda").manual_seed(7)
    cfg = adapter.GaussianAdapterCfg(0.5, 15.0, deg)
    plain, fused = adapter.UnifiedGaussianAdapter(cfg).cuda(), adapter.UnifiedGaussianAdapter(cfg, fuse_into_decoder=True).cuda()
    C = 7 + 3 * K
    head = torch.randn(2, 900, C + 1, device="cuda", generator=gen)
    head[..., 1:4] = head[..., 1:4] * 3.0 + 6.0                  # scales of a few per cent of the scene (some at the 0.3 clamp)
    head[..., 8:] *= 30.0                                          # (the mask scales the higher bands down by 40 .. 2,560)
    dec = spf.get_decoder(spf.DecoderSplattingCUDACfg("splatting_cuda", [0.1, 0.2, 0.3], True, True, True)).cuda()
    dec.auto_plan = None
    dec.sh_band4 = deg == 4
    w = torch.rand(2, views, 3, 64, 48, device="cuda", generator=gen)

    def run(ad, raw_leaf, raw_in):
        leaves = {"means": b.means.clone().requires_grad_(True), "opacities": b.opacities.clone().requires_grad_(True),
                  "extrinsics": b.extrinsics.clone().requires_grad_(True)}
        g = ad(leaves["means"], leaves["opacities"], raw_in, with_covariances=False)
        out, alpha, radii = dec.render(spf.Gaussians(g.means, None, g.rotations, g.scales, g.harmonics, g.opacities, raw=g.raw),
                                       leaves["extrinsics"], b.intrinsics, b.near, b.far, b.image_shape)
        ((out.color * w).sum() + 0.1 * (out.depth * w[:, :, 0]).sum() + 0.1 * (alpha * w[:, :, :1]).sum()).backward()
        return out, alpha, radii, {n: t.grad for n, t in leaves.items()}, raw_leaf.grad

    if strided:
        h1, h2 = head.clone().requires_grad_(True), head.clone().requires_grad_(True)
        a = run(plain, h1, h1[..., 1:].contiguous())
        f = run(fused, h2, h2[..., 1:])                           # a VIEW: row stride 7 + 3K + 1
        ga, gf = a[4][..., 1:], f[4][..., 1:]
        assert float(f[4][..., 0].abs().max()) == 0.0
    else:
        r1, r2 = head[..., 1:].contiguous().requires_grad_(True), head[..., 1:].contiguous().requires_grad_(True)
        a, f = run(plain, r1, r1), run(fused, r2, r2)
        ga, gf = a[4], f[4]
    from tests import util
    for got, want, what in ((f[0].color, a[0].color, "color"), (f[0].depth, a[0].depth, "depth"), (f[1], a[1], "alpha")):
        assert util.rel_linf(got, want) < 2e-6, (what, util.rel_linf(got, want))
    assert torch.equal(f[2], a[2])                                   # radii
    for n in a[3]:
        assert util.rel_linf(f[3][n], a[3][n]) < 2e-5, (n, util.rel_linf(f[3][n], a[3][n]))
    assert util.rel_linf(gf, ga) < 2e-5, util.rel_linf(gf, ga)
    assert util.rel_linf(gf[..., :7], ga[..., :7]) < 2e-5 and util.rel_linf(gf[..., 7:], ga[..., 7:]) < 2e-5
    assert float(ga[..., :7].abs().max()) > 0 and float(ga[..., 7:].abs().max()) > 0
    # the other consumers of a fused-mode Gaussians get the standard fields on request
    g = fused(b.means, b.opacities, head[..., 1:], with_covariances=False)
    m, p_ = adapter.materialize(g), plain(b.means, b.opacities, head[..., 1:].contiguous(), with_covariances=False)
    assert g.scales is None and torch.equal(m.scales, p_.scales) and torch.equal(m.harmonics, p_.harmonics)


def test_adapter_fused_into_the_decoder_against_the_oracle(hip_lib):
    """The fused path held against the ORACLE, not against the product's other path: raw rows -> a float64 torch
    restatement of UnifiedGaussianAdapter.forward (gaussian_adapter.py:122-150: 0.001 softplus clamped at 0.3, q / (|q| +
    eps), raw x sh_mask) -> oracle/splat_ref.py, its gradients chained back to the raw channels by autograd; the usual
    gates (RGB 1e-4, gradients 1e-3 of scale and 1e-2 per element, radii exact, knife-edge pixels masked on both sides)."""
    import spfsplatv2_amd as spf
    from spfsplatv2_amd import adapter, synthetic as syn
    K, S, V, G = 25, 2, 2, 1200
    batch = syn.make_batch("TEST", S, V, seed=77, s_mult=10.0, G=G, K=K, image_hw=(64, 48))
    gen = torch.Generator().manual_seed(78)
    raw = torch.randn(S, G, 7 + 3 * K, generator=gen)
    raw[..., :3] = raw[..., :3] * 8.0 + 60.0                    # scales of ~0.06 scene units (softplus' x > 20 branch) ...
    raw[:, :G // 3, :3] = torch.randn(S, G // 3, 3, generator=gen) * 2.0 + 2.0      # ... a third of them tiny (x < 20) ...
    raw[:, -10:, :3] = 400.0                                    # ... and ten at the 0.3 clamp (no gradient through it)
    raw[..., 7:] *= torch.cat((torch.ones(1), torch.full((3,), 40.0), torch.full((5,), 160.0), torch.full((7,), 640.0),
                               torch.full((9,), 2560.0))).repeat(3) * 0.5      # (so that the masked bands all matter)
    cfg = adapter.GaussianAdapterCfg(0.5, 15.0, 4)
    fused = adapter.UnifiedGaussianAdapter(cfg, fuse_into_decoder=True).cuda()
    mask64, eps = fused.sh_mask.double().cpu(), 1e-8

    def adapter64(r):                                          # gaussian_adapter.py:122-150, float64
        scales = (0.001 * torch.nn.functional.softplus(r[..., :3])).clamp_max(0.3)
        rot = r[..., 3:7] / (r[..., 3:7].norm(dim=-1, keepdim=True) + eps)
        return scales, rot, r[..., 7:].reshape(*r.shape[:-1], 3, K) * mask64

    with torch.no_grad():
        sc, ro, sh = adapter64(raw.double())
    batch.scales, batch.rotations, batch.harmonics = sc.float(), ro.float(), sh.float()
    ref = util.run_oracle(batch, torch.float64, background=(0.1, 0.2, 0.3), mask_fragile=True, band4=False)
    r64 = raw.double().requires_grad_(True)
    (g_raw,) = torch.autograd.grad(adapter64(r64), r64, (ref["grads"]["scales"].double(), ref["grads"]["rotations"].double(),
                                                        ref["grads"]["harmonics"].double()))
    # ---- the product: raw rows straight into the decoder ----
    bd = batch.to("cuda")
    leaves = {n: getattr(bd, n).clone().requires_grad_(True) for n in ("means", "opacities", "extrinsics")}
    raw_d = raw.cuda().requires_grad_(True)
    dec = util.product_decoder(background=(0.1, 0.2, 0.3), band4=False)
    g = fused(leaves["means"], leaves["opacities"], raw_d, with_covariances=False)
    out, alpha, radii = dec.render(spf.Gaussians(g.means, None, None, None, None, g.opacities, raw=g.raw),
                                   leaves["extrinsics"], bd.intrinsics, bd.near, bd.far, bd.image_shape)
    wd, wa = util.loss_weights(batch)
    util.scalar_loss(out.color, out.depth, alpha, bd.target, wd.cuda(), wa.cuda(), ref["pixel_mask"].cuda()).backward()
    prod = dict(color=out.color.detach().cpu(), depth=out.depth.detach().cpu(), alpha=alpha.detach().cpu(), radii=radii.cpu(),
                grads={**{n: t.grad.cpu() for n, t in leaves.items()}, "raw": raw_d.grad.cpu()})
    ref2 = dict(ref, grads={**{n: ref["grads"][n] for n in leaves}, "raw": g_raw})
    rep = util.compare(prod, ref2)
    assert not rep["fails"], rep
    assert rep["g_raw"] <= 1e-3 and rep["gel_raw"] <= 1e-2, rep
    assert float(g_raw[..., :3].abs().max()) > 0 and float(g_raw[..., 3:7].abs().max()) > 0 and float(g_raw[..., 7:].abs().max()) > 0
