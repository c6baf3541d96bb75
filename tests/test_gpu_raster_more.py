"""More GPU parity: long tile lists (every sort size class), colours-precomputed and orthographic entry points,
gradient switches, degenerate inputs, and size-independent properties at BASELINE's full C2 size."""
import pytest
import torch

from spfsplatv2_amd import synthetic as syn
from tests import util

pytestmark = pytest.mark.gpu


# flagged-pixel budgets of the long-list cases and the tolerance of their UNMASKED gradients (measured: see
# profiles/r04_parity_reports.jsonl, cases long_lists_G*)
# Round 3 allowed 25 % / 50 % of these images to be masked; with round 4's windows the measured fractions are 0.1 % (3,200 -
# 12,000 Gaussians), 1.5 % / 2.9 % (50,000 / 110,000) and 5.5 % (131,072 in ONE tile) -- and the unmasked gradients of the
# two register-sort classes agree to 5e-6 (no pixel flips at all there), so they are gated at the ordinary 1e-3.
LONG_LIST_FRAGILE_CAP = 0.04
LONG_LIST_FRAGILE_CAP_ONE_TILE = 0.08
UNMASKED_GRAD_TOL = 1e-3


@pytest.mark.parametrize("G,expect_min_list,hw", [(3200, 513, None), (6500, 1025, None), (12000, 2049, None),
                                                   (50000, 8193, None), (110000, 16385, None),
                                                   (131072, 32769, (16, 16))])
def test_long_tile_lists_every_sort_class(hip_lib, G, expect_min_list, hw):
    """4 tiles, thousands of Gaussians each: the block-per-tile register sorts (512, 1024] and (1024, 2048], the LDS
    sort classes (2048, 8192], (8192, 16384] and, above 16384 entries, the
    chunked sort (16384-entry chunks in LDS, merges across chunks through global memory: one level at 110k, two at
    131k Gaussians)."""
    batch = syn.make_batch("TESTBIG", 1, 1, seed=21, s_mult=1.0, G=G, image_hw=hw)     # (16x16: everything in ONE tile)
    batch.opacities = batch.opacities * 0.03        # keep transmittance alive deep into the lists
    # Knife-edge pixels are excluded from the RGB gate AND switched off in the loss of both sides, as in every other
    # parity test: every pixel is reached by thousands of entries here, so a good part of the image sits next to some
    # alpha / stop threshold and a single flipped contribution is a 1e-3 gradient difference by itself.  The gradients
    # of the loss over ALL pixels are compared too (`gall_*` in the report) and, for the two register-sort classes,
    # GATED at UNMASKED_GRAD_TOL -- a regression of the long-list backward cannot hide behind the mask.
    small = G <= 6500
    ref = util.run_oracle(batch, torch.float64, mask_fragile=True, unmasked_too=small)
    prod = util.run_product(batch, pixel_mask=ref["pixel_mask"], unmasked_too=small)
    assert prod["stats"]["max_tile_list"] >= expect_min_list, prod["stats"]
    rep = util.compare(prod, ref, max_fragile_frac=LONG_LIST_FRAGILE_CAP if hw is None else LONG_LIST_FRAGILE_CAP_ONE_TILE)
    from tests.test_gpu_raster import _report
    _report(f"long_lists_G{G}", rep)
    assert not rep["fails"], rep
    if small:
        worst = max(v for k, v in rep.items() if k.startswith("gall_"))
        assert worst < UNMASKED_GRAD_TOL, rep


def test_render_cuda_colors_precomp_matches_oracle(hip_lib):
    """`render_cuda(..., use_sh=False)` -> colors_precomp path (cuda_splatting.py:131-132)."""
    import spfsplatv2_amd as spf
    from oracle import glue_ref, splat_ref
    b = syn.make_batch("TEST", 3, 1, seed=31, s_mult=12.0, G=900, K=1, image_hw=(48, 80))
    b.harmonics = b.harmonics.abs()                     # colours used as given
    bg = torch.rand(3, 3, generator=torch.Generator().manual_seed(1))
    dev = "cuda"
    harm = b.harmonics.to(dev).requires_grad_(True)
    img, dep = spf.render_cuda(b.extrinsics[:, 0].to(dev), b.intrinsics[:, 0].to(dev), b.near[:, 0].to(dev),
                               b.far[:, 0].to(dev), b.image_shape, bg.to(dev), b.means.to(dev), b.covariances.to(dev),
                               harm, b.opacities.to(dev), b.rotations.to(dev), b.scales.to(dev),
                               scale_invariant=True, use_sh=False, enable_cov_grad=True, enable_sh_grad=True)
    assert img.shape == (3, 3, 48, 80) and dep.shape == (3, 1, 48, 80)
    img.sum().backward()
    args = glue_ref.callsite_args(b.extrinsics[:, 0], b.intrinsics[:, 0], b.near[:, 0], b.far[:, 0], b.image_shape,
                                  bg, b.means, b.harmonics, b.opacities, b.rotations, b.scales, use_sh=False)
    for i, a in enumerate(args):
        col = a["colors_precomp"].double().requires_grad_(True)
        oi, od, oa, _, frag = splat_ref.rasterize(
            a["means3D"].double(), a["scales"].double(), a["rotations"].double(), a["opacities"].double(), None,
            col, a["viewmatrix"].double(), a["projmatrix"].double(), a["bg"].double(), a["tanfovx"], a["tanfovy"],
            48, 80, 0, want_fragile=True)
        ok = ~frag
        assert float(((img[i].detach().cpu().double() - oi).abs() * ok).max()) < 1e-4
        assert float(((dep[i].detach().cpu().double() - od).abs() * ok).max()) < 1e-4 * float(od.max())
        oi.sum().backward()
        assert util.rel_linf(harm.grad[i, :, :, 0], col.grad) < 1e-3


def test_orthographic_matches_oracle(hip_lib):
    """`render_cuda_orthographic` (cuda_splatting.py:146-255; caller validate_in_the_wild.py:326-341) at north_star's
    tolerance: RGB 1e-4, plus depth and alpha through the same camera tensors.  tan(fov/2) ~ 9e-4 puts the camera
    ~10^3 units back; both sides build the view matrix as move_back^-1 @ extrinsics^-1 (see
    `spfsplatv2_amd.orthographic_camera`), so no 10^3-sized cancellation is left in the pixel centres."""
    import spfsplatv2_amd as spf
    from oracle import glue_ref, splat_ref
    b = syn.make_batch("TEST", 2, 1, seed=32, s_mult=25.0, G=1200, K=4, image_hw=(64, 64))
    dev = "cuda"
    width, height = torch.tensor([6.0, 9.0]), torch.tensor([6.0, 7.0])
    near, far = torch.tensor([0.5, 0.5]), torch.tensor([60.0, 60.0])
    bg = torch.tensor([[0.1, 0.2, 0.3], [0.0, 0.0, 0.0]])
    out = spf.render_cuda_orthographic(b.extrinsics[:, 0].to(dev), width.to(dev), height.to(dev), near.to(dev),
                                       far.to(dev), (40, 56), bg.to(dev), b.means.to(dev),
                                       b.covariances.to(dev), b.harmonics.to(dev), b.opacities.to(dev),
                                       b.rotations.to(dev), b.scales.to(dev), fov_degrees=0.1)
    # the same call one level down, for the outputs the wrapper drops
    view, proj, tanfov = spf.orthographic_camera(b.extrinsics[:, 0].to(dev), width.to(dev), height.to(dev),
                                                 near.to(dev), far.to(dev), 0.1)
    img2, dep, alp, radii = spf.rasterize_batch(b.means.to(dev), b.scales.to(dev), b.rotations.to(dev),
                                                b.opacities.to(dev), b.harmonics.transpose(-1, -2).contiguous().to(dev),
                                                None, view[:, None], proj[:, None], tanfov, bg[:, None].to(dev), 40, 56, 1)
    assert torch.equal(img2[:, 0], out)
    args = glue_ref.orthographic_callsite_args(b.extrinsics[:, 0], width, height, near, far, (40, 56), bg, b.means,
                                               b.harmonics, b.opacities, b.rotations, b.scales)
    for i, a in enumerate(args):
        oi, od, oa, orad, frag, rfrag = splat_ref.rasterize(
            a["means3D"].double(), a["scales"].double(), a["rotations"].double(), a["opacities"].double(),
            a["shs"].double(), None, a["viewmatrix"].double(), a["projmatrix"].double(), a["bg"].double(),
            a["tanfovx"], a["tanfovy"], 40, 56, a["sh_degree"], want_fragile=True, want_radii_fragile=True)
        # (the camera sits 3,400 - 5,200 units back, where float32 resolves depth to 2.4e-4 - 4.9e-4: under splats this
        #  wide a few percent of the pixels have two contributors whose ORDER float32 cannot tell -- flagged)
        assert float(oa.max()) > 0.05 and float(frag.float().mean()) < 0.05           # something is actually visible
        ok = ~frag
        assert float(((out[i].cpu().double() - oi).abs() * ok).max()) < 1e-4
        assert float(((alp[i, 0].cpu().double() - oa).abs() * ok).max()) < 1e-4
        assert float(((dep[i, 0].cpu().double() - od).abs() * ok).max()) < 1e-4 * float(od.max())
        assert int(((radii[i, 0].cpu() != orad) & ~rfrag).sum()) == 0


def test_grad_switches(hip_lib):
    """enable_cov_grad=False -> no gradient to scales/rotations; enable_sh_grad=False -> none to SH
    (SURVEY.md Appendix B #12); everything else unchanged."""
    import spfsplatv2_amd as spf
    b = syn.make_batch("TEST", 1, 2, seed=33, s_mult=10.0, G=600, K=4, image_hw=(48, 48)).to("cuda")

    def run(cov, sh):
        leaves = {n: getattr(b, n).clone().requires_grad_(True) for n in util.GRAD_NAMES}
        color, _, _ = spf.render_views(leaves["extrinsics"], b.intrinsics, b.near, b.far, b.image_shape,
                                       torch.zeros(3, device="cuda"), leaves["means"], leaves["harmonics"],
                                       leaves["opacities"], leaves["rotations"], leaves["scales"],
                                       enable_cov_grad=cov, enable_sh_grad=sh)
        ((color - b.target) ** 2).mean().backward()
        return {n: leaves[n].grad for n in util.GRAD_NAMES}

    full, off = run(True, True), run(False, False)
    assert off["scales"] is None and off["rotations"] is None and off["harmonics"] is None
    for n in ("means", "opacities", "extrinsics"):
        assert util.rel_linf(off[n], full[n]) < 1e-4


def test_degenerate_inputs(hip_lib):
    import spfsplatv2_amd as spf
    dev = "cuda"
    b = syn.make_batch("C1", 1, 1, seed=34, s_mult=30.0)
    # (a) everything behind the camera: D = 0, image = background exactly
    means = b.means.clone()
    means[..., 2] = -5.0
    bg = torch.tensor([0.25, 0.5, 0.75], device=dev)
    color, depth, alpha = spf.render_views(b.extrinsics.to(dev), b.intrinsics.to(dev), b.near.to(dev), b.far.to(dev),
                                           (64, 64), bg, means.to(dev).requires_grad_(True), b.harmonics.to(dev),
                                           b.opacities.to(dev), b.rotations.to(dev), b.scales.to(dev))
    assert spf.last_forward_stats()["num_pairs"] == 0
    assert torch.equal(color[0, 0], bg[:, None, None].expand(3, 64, 64)) and float(alpha.abs().max()) == 0.0
    color.sum().backward()                               # backward with zero pairs must not fault
    # (b) one Gaussian, image smaller than a tile, odd size
    one = syn.make_batch("C1", 1, 1, seed=35, s_mult=200.0, G=1, image_hw=(7, 11))
    prod = util.run_product(one)
    ref = util.run_oracle(one, torch.float64)
    rep = util.compare(prod, ref)
    assert not rep["fails"], rep
    # (c) NaN / inf parameters are culled, the rest renders
    bad = syn.make_batch("C1", 1, 1, seed=36, s_mult=30.0)
    bad.means[0, :5] = float("nan")
    bad.scales[0, 5:9] = float("inf")
    c2, _, _ = spf.render_views(bad.extrinsics.to(dev), bad.intrinsics.to(dev), bad.near.to(dev), bad.far.to(dev),
                                (64, 64), bg, bad.means.to(dev), bad.harmonics.to(dev), bad.opacities.to(dev),
                                bad.rotations.to(dev), bad.scales.to(dev))
    assert bool(torch.isfinite(c2).all())


@pytest.fixture(scope="module")
def c2_batch():
    return syn.make_batch("C2", 2, 2, seed=1000).to("cuda")


def _render(b, means=None, colors=None, opac=None, bg=(0.0, 0.0, 0.0), harm=None):
    import spfsplatv2_amd as spf
    harm = b.harmonics if harm is None else harm
    return spf.render_views(b.extrinsics, b.intrinsics, b.near, b.far, b.image_shape,
                            torch.tensor(bg, device="cuda"), b.means if means is None else means, harm,
                            b.opacities if opac is None else opac, b.rotations, b.scales,
                            use_sh=colors is None, enable_cov_grad=True, enable_sh_grad=True) \
        if colors is None else \
        spf.render_views(b.extrinsics, b.intrinsics, b.near, b.far, b.image_shape, torch.tensor(bg, device="cuda"),
                         b.means if means is None else means, colors, b.opacities if opac is None else opac,
                         b.rotations, b.scales, use_sh=False, enable_cov_grad=True, enable_sh_grad=True)


def test_full_size_properties(hip_lib, c2_batch):
    """BASELINE config 2 size (65,536 Gaussians, 256x256): properties that need no oracle."""
    b = c2_batch
    img, dep, alp = _render(b)
    assert bool(torch.isfinite(img).all()) and bool(torch.isfinite(dep).all())
    assert float(alp.min()) >= 0.0 and float(alp.max()) <= 1.0 and float(alp.mean()) > 0.3
    img2, dep2, alp2 = _render(b)
    assert torch.equal(img, img2) and torch.equal(dep, dep2)                      # forward is bit-stable
    # background enters as T_final * bg = (1 - alpha) * bg
    bgc = (0.3, 0.6, 0.9)
    img_bg, _, _ = _render(b, bg=bgc)
    want = img + (1 - alp) * torch.tensor(bgc, device="cuda")[None, None, :, None, None]
    assert float((img_bg - want).abs().max()) < 2e-6
    # colours given directly: the image is linear in them (blend weights depend on geometry/opacity only)
    gen = torch.Generator().manual_seed(5)
    c1 = torch.rand(b.harmonics.shape, generator=gen).cuda()
    c2 = torch.rand(b.harmonics.shape, generator=gen).cuda()
    i1, _, _ = _render(b, colors=c1)
    i2, _, _ = _render(b, colors=c2)
    i12, _, _ = _render(b, colors=c1 + c2)
    assert float((i12 - i1 - i2).abs().max()) < 5e-6
    # order of the Gaussians in memory does not matter (depth ties aside)
    perm = torch.randperm(b.means.shape[1], generator=gen).cuda()
    bp = syn.Batch(**{**b.__dict__, "means": b.means[:, perm], "scales": b.scales[:, perm],
                      "rotations": b.rotations[:, perm], "opacities": b.opacities[:, perm],
                      "harmonics": b.harmonics[:, perm]})
    ip, _, _ = _render(bp)
    assert float((ip - img).abs().max()) < 1e-5


def test_full_size_gradient_checksum(hip_lib, c2_batch):
    """With loss = sum(image), bg = 0 and colours given directly: d loss / d colour_g = sum over pixels of the blend
    weight, so sum_g dL/dcolour_g[c] = sum_pixels (1 - T_final) = sum(alpha) for every channel -- a checksum over
    D = 350k (Gaussian, tile) pairs that exercises binning, sort order, LDS accumulation and the pair records."""
    b = c2_batch
    col = torch.rand(b.harmonics.shape, generator=torch.Generator().manual_seed(6)).cuda().requires_grad_(True)
    img, _, alp = _render(b, colors=col)
    img.sum().backward()
    per_scene = col.grad[..., 0].sum(dim=1)                      # [S,3]
    want = alp.sum(dim=(1, 2, 3, 4))                             # [S]
    assert float(((per_scene - want[:, None]).abs() / want[:, None]).max()) < 1e-4
    # opacity gradient of the same loss is finite and not identically zero
    op = b.opacities.clone().requires_grad_(True)
    i2, _, _ = _render(b, opac=op)
    i2.sum().backward()
    assert bool(torch.isfinite(op.grad).all()) and float(op.grad.abs().max()) > 0


def test_decoder_module_with_a_planned_budget(hip_lib):
    """DecoderSplattingCUDA: one exact call, then `decoder.max_pairs = plan_pair_budget(...)` -- same pixels, no
    read-back, the plan verified on the device."""
    import spfsplatv2_amd as spf
    from spfsplatv2_amd import decoder as dec
    batch = syn.make_batch("TEST", 2, 3, seed=31, s_mult=3.0, G=3000, K=4, image_hw=(96, 80)).to("cuda")
    d = dec.get_decoder(dec.DecoderSplattingCUDACfg(name="splatting_cuda", background_color=[0.1, 0.2, 0.3],
                                                    make_scale_invariant=True, enable_cov_grad=True,
                                                    enable_sh_grad=True)).cuda()
    g = dec.Gaussians(batch.means, batch.covariances, batch.rotations, batch.scales, batch.harmonics, batch.opacities)
    exact = d(g, batch.extrinsics, batch.intrinsics, batch.near, batch.far, batch.image_shape)
    d.max_pairs = spf.plan_pair_budget(check="deferred")
    planned = d(g, batch.extrinsics, batch.intrinsics, batch.near, batch.far, batch.image_shape)
    assert spf.last_plan_flags() == 0
    assert torch.equal(exact.color, planned.color) and torch.equal(exact.depth, planned.depth)


def test_rendered_norm_and_extra_are_opt_in_blends(hip_lib):
    """Positions 2 and 5 of the rasterizer's 6-tuple (cuda_splatting.py:128): None by default (the reference discards
    them), alpha-blended per-Gaussian attributes on request -- checked against the oracle's blend of the same attributes,
    and against geometry: a disc facing the camera has normal (0, 0, -1) in view space."""
    import spfsplatv2_amd as spf
    from oracle import glue_ref, splat_ref
    from spfsplatv2_amd.rasterizer import gaussian_normals
    batch = syn.make_batch("TEST", 1, 1, seed=9, s_mult=12.0, G=800, K=4, image_hw=(64, 64))
    a = glue_ref.callsite_args(batch.extrinsics[:, 0], batch.intrinsics[:, 0], batch.near[:, 0], batch.far[:, 0],
                               batch.image_shape, torch.zeros(1, 3), batch.means, batch.harmonics, batch.opacities,
                               batch.rotations, batch.scales)[0]
    dev = "cuda"
    t = lambda x: x.to(dev)
    kw = dict(image_height=64, image_width=64, tanfovx=a["tanfovx"], tanfovy=a["tanfovy"], bg=t(a["bg"]),
              scale_modifier=1.0, projmatrix=t(a["projmatrix"]), sh_degree=a["sh_degree"])
    call = dict(means3D=t(a["means3D"]), means2D=None, shs=t(a["shs"]), colors_precomp=None, opacities=t(a["opacities"]),
                scales=t(a["scales"]), rotations=t(a["rotations"]), viewmatrix=t(a["viewmatrix"]))
    plain = spf.GaussianRasterizer(spf.GaussianRasterizationSettings(**kw))(**call)
    assert plain[2] is None and plain[5] is None
    gen = torch.Generator().manual_seed(3)
    attrs = torch.randn(800, 5, generator=gen)
    scales = t(a["scales"]).clone().requires_grad_(True)
    call["scales"] = scales
    out = spf.GaussianRasterizer(spf.GaussianRasterizationSettings(**kw, render_norm=True))(**call, extra_attrs=t(attrs))
    norm, extra = out[2], out[5]
    assert norm.shape == (3, 64, 64) and extra.shape == (5, 64, 64) and torch.equal(out[0], plain[0])
    n_cpu = gaussian_normals(a["means3D"], a["scales"], a["rotations"], a["viewmatrix"])
    assert float((n_cpu.norm(dim=-1) - 1).abs().max()) < 1e-5
    for attr, got in ((n_cpu, norm), (attrs[:, :3], extra[:3]), (torch.cat([attrs[:, 3:], torch.zeros(800, 1)], 1), extra[3:])):
        want, _, _, _, frag = splat_ref.rasterize(
            a["means3D"].double(), a["scales"].double(), a["rotations"].double(), a["opacities"].double(), None,
            attr.double(), a["viewmatrix"].double(), a["projmatrix"].double(), torch.zeros(3).double(), a["tanfovx"],
            a["tanfovy"], 64, 64, 0, want_fragile=True)
        n = got.shape[0]
        assert float(((got.detach().cpu().double() - want[:n]).abs() * ~frag).max()) < 1e-4
    norm.square().sum().backward()                       # differentiable like any colour (here: through argmin's axis)
    assert scales.grad is not None and bool(torch.isfinite(scales.grad).all())
    # one opaque disc, thin along the view axis, in front of an identity camera: its normal faces the camera
    one = spf.GaussianRasterizer(spf.GaussianRasterizationSettings(
        **{**kw, "projmatrix": t(a["projmatrix"])}, render_norm=True))(
        means3D=torch.tensor([[0.0, 0.0, 5.0]], device=dev), means2D=None, shs=None,
        colors_precomp=torch.ones(1, 3, device=dev), opacities=torch.tensor([[0.9]], device=dev),
        scales=torch.tensor([[0.5, 0.5, 0.01]], device=dev), rotations=torch.tensor([[1.0, 0, 0, 0]], device=dev),
        viewmatrix=torch.eye(4, device=dev))
    n, al = one[2], one[3]
    assert float(al.max()) > 0.5
    assert float((n[2] + al[0]).abs().max()) < 1e-5 and float(n[:2].abs().max()) < 1e-6      # (0, 0, -1) * alpha


@pytest.mark.parametrize("hw", [(1024, 1024), (1040, 1072)], ids=["4096_tiles_lds", "4355_tiles_global_atomics"])
def test_large_images_both_tile_counter_paths(hip_lib, hw):
    """1024x1024 is what the reference's paper script renders (validate_in_the_wild.py:304): 4,096 tiles, the most the
    per-block LDS histograms of the projection / binning kernels hold; one tile row more and they fall back to global
    atomics.  Forward against the oracle at full size (its autograd pass over 4,000 tiles takes minutes: the gradient
    parity of the fall-back path is `test_global_atomic_tile_counters_with_gradients`)."""
    batch = syn.make_batch("TEST", 1, 1, seed=61, s_mult=40.0, G=2500, K=4, image_hw=hw)
    ref = util.run_oracle(batch, torch.float64, background=(0.1, 0.0, 0.2), with_grads=False)
    prod = util.run_product(batch, background=(0.1, 0.0, 0.2))
    rep = util.compare(prod, ref)
    assert not rep["fails"], rep
    assert prod["stats"]["tiles"] == ((hw[0] + 15) // 16) * ((hw[1] + 15) // 16) and prod["stats"]["num_pairs"] > 2000
    assert all(bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0 for g in prod["grads"].values())


def test_global_atomic_tile_counters_with_gradients(hip_lib, monkeypatch):
    """The > 4096-tile path of the projection and binning kernels (no LDS histograms) forced on a small image
    (`SPF_MAX_LDS_TILES`, read at every launch): full parity incl. gradients, and bit-equal images with the LDS path."""
    batch = syn.make_batch("TEST", 2, 2, seed=62, s_mult=8.0, G=1500, K=4, image_hw=(80, 112))
    ref = util.run_oracle(batch, torch.float64, mask_fragile=True)
    lds = util.run_product(batch, pixel_mask=ref["pixel_mask"])
    monkeypatch.setenv("SPF_MAX_LDS_TILES", "1")
    glob = util.run_product(batch, pixel_mask=ref["pixel_mask"])
    rep = util.compare(glob, ref)
    assert not rep["fails"], rep
    assert torch.equal(glob["color"], lds["color"]) and torch.equal(glob["radii"], lds["radii"])


def test_many_renders_both_scan_kernels_agree(hip_lib):
    """More than 256 renders in one call fall back from the one-block-per-render tile scan to the single-block scan:
    320 views at once (single-block) are bit-equal to the same views rendered as 2 x 160 (one block per render), and
    the gradients agree."""
    import spfsplatv2_amd as spf
    b = syn.make_batch("TEST", 1, 320, seed=71, s_mult=20.0, G=400, K=4, image_hw=(32, 48)).to("cuda")

    def run(vs):
        means = b.means.clone().requires_grad_(True)
        color, depth, alpha = spf.render_views(b.extrinsics[:, vs], b.intrinsics[:, vs], b.near[:, vs], b.far[:, vs],
                                               b.image_shape, torch.zeros(3, device="cuda"), means, b.harmonics,
                                               b.opacities, b.rotations, b.scales, enable_cov_grad=True,
                                               enable_sh_grad=True)
        (color * b.target[:, vs]).sum().backward()
        return color, depth, means.grad

    c_all, d_all, g_all = run(slice(0, 320))
    c_a, d_a, g_a = run(slice(0, 160))
    c_b, d_b, g_b = run(slice(160, 320))
    assert torch.equal(c_all, torch.cat([c_a, c_b], dim=1)) and torch.equal(d_all, torch.cat([d_a, d_b], dim=1))
    assert float(c_all.abs().max()) > 0.1
    assert util.rel_linf(g_all, g_a + g_b) < 1e-5


def test_both_sort_families_give_the_same_lists(hip_lib, monkeypatch):
    """Lists of 513..2048 entries are sorted by one wave per tile (16 / 32 keys per lane) when there are many tiles and by
    one 256-thread block per tile when there are few (`SPF_SORT_BLOCKS` pins the family): the sorted lists are unique
    (keys are), so images and gradients must be bit-equal -- and one of the two runs is gated against the oracle by the
    long-list cases above."""
    outs = []
    for fam in ("0", "1"):
        monkeypatch.setenv("SPF_SORT_BLOCKS", fam)
        per = []
        for G in (3200, 6500):
            batch = syn.make_batch("TESTBIG", 1, 1, seed=21, s_mult=1.0, G=G)
            batch.opacities = batch.opacities * 0.03
            p = util.run_product(batch)
            assert p["stats"]["max_tile_list"] > 512
            per.append(p)
        outs.append(per)
    for a, b in zip(*outs):
        assert torch.equal(a["color"], b["color"]) and torch.equal(a["depth"], b["depth"])
        for n in util.GRAD_NAMES:
            assert torch.equal(a["grads"][n], b["grads"][n]), n


def test_long_list_class_inside_the_mixed_sort_launch_gives_the_same_lists(hip_lib, monkeypatch):
    """Calls of few tiles sort lists of 2,049..4,096 entries (sixteen keys per thread) in the same launch as the shorter
    classes (`spf_sort_tiles_mixed_kernel<true>`); `SPF_SORT_BIG_MIXED=0` gives that class a launch of its own, as before.
    Same networks on the same lists: bit-equal images and gradients, exact and planned."""
    import spfsplatv2_amd as spf
    batch, G = None, 9000
    for _ in range(8):                                   # (the longest list grows ~linearly with G: aim at 2,600)
        cand = syn.make_batch("TESTBIG", 1, 1, seed=23, s_mult=1.0, G=G)
        cand.opacities = cand.opacities * 0.03
        longest = util.run_product(cand, with_grads=False)["stats"]["max_tile_list"]
        if 2048 < longest <= 3400:
            batch = cand
            break
        G = max(int(G * 2600 / max(longest, 1)), 64)
    assert batch is not None, ("no batch with a longest list of 2,049..3,400 entries", G, longest)
    outs = []
    for big in ("0", "1"):
        monkeypatch.setenv("SPF_SORT_BIG_MIXED", big)
        exact = util.run_product(batch)
        planned = util.run_product(batch, max_pairs=spf.plan_pair_budget(exact["stats"], check="deferred"))
        assert spf.plan_flags(planned["decoder"].last_call) == 0
        outs.append((exact, planned))
    monkeypatch.delenv("SPF_SORT_BIG_MIXED")
    for a, b in zip(*outs):
        assert torch.equal(a["color"], b["color"]) and torch.equal(a["depth"], b["depth"])
        for n in util.GRAD_NAMES:
            assert torch.equal(a["grads"][n], b["grads"][n]), n


def test_two_views_per_binning_block_give_the_same_lists(hip_lib, monkeypatch):
    """Renders of many blocks bin two views of a scene per block (`spf_bin_pairs_views_kernel`, picked from G >= 262,144;
    `SPF_BIN_VIEWS` pins it): bins fill in another order, the sorted lists are the same -- bit-equal images and gradients,
    for an odd number of views (the last block of a scene holds one) and for several scenes."""
    outs = []
    for vb in ("1", "2"):
        monkeypatch.setenv("SPF_BIN_VIEWS", vb)
        per = []
        for S, V, G in ((1, 3, 3000), (2, 2, 1500), (1, 5, 700)):
            batch = syn.make_batch("TEST", S, V, seed=31, s_mult=2.0, G=G, K=4, image_hw=(80, 72))
            per.append(util.run_product(batch))
        outs.append(per)
    for a, b in zip(*outs):
        assert a["stats"]["num_pairs"] == b["stats"]["num_pairs"]
        assert torch.equal(a["color"], b["color"]) and torch.equal(a["depth"], b["depth"])
        for n in util.GRAD_NAMES:
            assert torch.equal(a["grads"][n], b["grads"][n]), n


def test_wave_pair_sort_gives_the_same_lists_as_single_waves(hip_lib, monkeypatch):
    """Many-tiles family, lists of 257 .. 1024 entries: two waves share a list (`spf_sort_tiles_pair_kernel`, the default)
    or one wave takes it whole (`SPF_SORT_SINGLE=1`): same unique order, so bit-equal images and gradients -- for both
    pair classes (4 and 8 keys per thread)."""
    monkeypatch.setenv("SPF_SORT_BLOCKS", "0")
    outs, seen = [], set()
    for single in ("", "1"):
        if single:
            monkeypatch.setenv("SPF_SORT_SINGLE", single)
        else:
            monkeypatch.delenv("SPF_SORT_SINGLE", raising=False)
        per = []
        for G in (800, 1600, 1800, 2200):                        # longest lists: 400, 466, 721, 989
            batch = syn.make_batch("TESTBIG", 1, 1, seed=21, s_mult=1.0, G=G)
            batch.opacities = batch.opacities * 0.03
            p = util.run_product(batch)
            m = p["stats"]["max_tile_list"]
            seen.add("4" if 256 < m <= 512 else ("8" if 512 < m <= 1024 else "-"))
            per.append(p)
        outs.append(per)
    assert {"4", "8"} <= seen, seen
    for a, b in zip(*outs):
        assert torch.equal(a["color"], b["color"]) and torch.equal(a["depth"], b["depth"])
        for n in util.GRAD_NAMES:
            assert torch.equal(a["grads"][n], b["grads"][n]), n


@pytest.mark.parametrize("crowd", [0, 3000], ids=["rows_kernel", "lists_kernel"])
def test_far_off_centre_anisotropic_splat_gradients(hip_lib, crowd):
    """A thin splat 0.09 units in front of the near plane whose centre projects ~490 px outside an 80x25 image (radius
    910 px; found by tools/fuzz_campaign.py, seed 2428).  Summing dL/dconic over the pixels and converting once per
    Gaussian -- the classic order -- cancels three ~1e-2 terms to ~1e-5 in float32: 1.3 % error on dL/dmeans.  The pair
    records carry dL/d(cov2D) formed per pixel from v = conic * offset instead (spf_common.h); both render backward
    forms are covered: alone the splat's tiles are dense (rows form), among `crowd` pixel-aligned small ones they
    stay sparse (lists kernel, whole-tile slot box)."""
    b = syn.make_batch("TEST", 1, 1, seed=77, s_mult=1.0, G=max(crowd, 1) + 1, K=4, image_hw=(80, 25))
    b.means[0, 0] = torch.tensor([-0.5659291744232178, -0.7119755148887634, 2.077361822128296])
    b.scales[0, 0] = torch.tensor([0.4121701121330261, 0.2872806489467621, 0.02158804051578045])
    b.rotations[0, 0] = torch.tensor([-0.5096940994262695, -0.12323477864265442, -0.7403966188430786, 0.4205211102962494])
    b.opacities[0, 0] = 0.26113107800483704
    b.harmonics[0, 0] = torch.tensor([[0.4860185384750366, -0.03290138393640518, 0.3309798538684845, 0.12275010347366333],
                                      [-0.44483447074890137, 0.0181439146399498, -0.4668569564819336, 0.2138572782278061],
                                      [0.5437580943107605, 0.336381733417511, 0.11369559913873672, 0.28525733947753906]])
    if crowd == 0:
        b.opacities[0, 1] = 0.0                       # (the one filler Gaussian: invisible)
    b.extrinsics[0, 0] = torch.tensor([[0.9993013739585876, -0.03478962555527687, 0.013653418980538845, 0.12748293578624725],
                                       [0.034742217510938644, 0.9993894696235657, 0.0036945438478142023, -0.04849812015891075],
                                       [-0.013773615472018719, -0.003217612626031041, 0.9998999834060669, 1.9743455648422241],
                                       [0.0, 0.0, 0.0, 1.0]])
    b.near[0, 0] = 0.17093364894390106
    ref = util.run_oracle(b, torch.float64, mask_fragile=True)
    prod = util.run_product(b, pixel_mask=ref["pixel_mask"])
    assert int(prod["radii"][0, 0, 0]) > 500
    st = prod["stats"]
    assert (st["dense_tiles"] == st["tiles"]) if crowd == 0 else (st["dense_tiles"] == 0), st
    rep = util.compare(prod, ref, max_fragile_frac=0.05)
    assert not rep["fails"], rep
    # the big splat's own gradient, against its own scale (the tensor-wide gate above is dominated by the crowd)
    for n in ("means", "scales", "rotations"):
        g, r = prod["grads"][n][0, 0].double(), ref["grads"][n][0, 0].double()
        assert float(r.abs().max()) > 0
        assert float((g - r).abs().max() / r.abs().max()) < 1e-3, (n, g.tolist(), r.tolist())
