"""Call-site contract: both the oracle glue and the PRODUCT glue reproduce exactly what the reference's own
Python hands to its rasterizer (fixtures captured by tests/golden/make_callsite_goldens.py)."""
import pytest
import torch

from oracle import glue_ref

TOL = 1e-6


def _close(a, b, tol=TOL):
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(b.abs().max()))
    assert float((a - b).abs().max()) <= tol * scale, float((a - b).abs().max())


def _check_call(got: dict, want: dict):
    s, k = want["settings"], want["kwargs"]
    assert got["image_height"] == s["image_height"] and got["image_width"] == s["image_width"]
    assert abs(got["tanfovx"] - s["tanfovx"]) <= TOL and abs(got["tanfovy"] - s["tanfovy"]) <= TOL
    assert got["sh_degree"] == s["sh_degree"] and got["scale_modifier"] == s["scale_modifier"]
    _close(got["bg"], s["bg"])
    _close(got["projmatrix"], s["projmatrix"])
    for name in ("means3D", "opacities", "scales", "rotations", "viewmatrix"):
        _close(got[name], k[name])
    for name in ("shs", "colors_precomp"):
        if k[name] is None:
            assert got[name] is None
        else:
            _close(got[name], k[name])


def test_camera_tables(golden_dir):
    from spfsplatv2_amd.decoder import get_fov, get_projection_matrix
    t = torch.load(golden_dir / "camera_tables.pt")
    for fov_fn, proj_fn in ((glue_ref.fov_from_intrinsics, glue_ref.projection_matrix),
                            (get_fov, get_projection_matrix)):
        fov = fov_fn(t["intrinsics"])
        _close(fov, t["fov"])
        _close(proj_fn(t["near"], t["far"], t["fov"][:, 0], t["fov"][:, 1]), t["proj"])


@pytest.mark.parametrize("tag", ["decoder_k4_si", "decoder_k25_nosi"])
def test_oracle_glue_matches_reference_callsite(golden_dir, tag):
    g = torch.load(golden_dir / f"callsite_{tag}.pt")
    i = g["inputs"]
    b, v = i["extrinsics"].shape[:2]
    rep = lambda t: t[:, None].expand(b, v, *t.shape[1:]).reshape(b * v, *t.shape[1:])
    bg = torch.tensor(i["background_color"])[None].expand(b * v, 3)
    got = glue_ref.callsite_args(i["extrinsics"].reshape(b * v, 4, 4), i["intrinsics"].reshape(b * v, 3, 3),
                                 i["near"].reshape(-1), i["far"].reshape(-1), i["image_shape"], bg, rep(i["means"]),
                                 rep(i["harmonics"]), rep(i["opacities"]), rep(i["rotations"]), rep(i["scales"]),
                                 scale_invariant=i["make_scale_invariant"])
    assert len(got) == len(g["calls"]) == b * v
    for a, w in zip(got, g["calls"]):
        _check_call(a, w)
    assert g["calls"][0]["viewmatrix_requires_grad"]          # pose gradients are load-bearing
    assert not g["calls"][0]["projmatrix_is_contiguous"]      # the reference passes a transposed view


def test_oracle_glue_precomp_and_orthographic(golden_dir):
    g = torch.load(golden_dir / "callsite_render_cuda.pt")
    i = g["inputs"]
    got = glue_ref.callsite_args(i["extrinsics"], i["intrinsics"], i["near"], i["far"], i["image_shape"], i["bg"],
                                 i["means"], i["harmonics"], i["opacities"], i["rotations"], i["scales"],
                                 scale_invariant=True, use_sh=False)
    for a, w in zip(got, g["calls_precomp"]):
        _check_call(a, w)
    got = glue_ref.orthographic_callsite_args(i["extrinsics"], i["ortho_width"], i["ortho_height"], i["near"],
                                              i["far"], i["ortho_image_shape"], torch.zeros(3, 3), i["means"],
                                              i["harmonics"], i["opacities"], i["rotations"], i["scales"])
    for a, w in zip(got, g["calls_ortho"]):
        _check_call(a, w)


class _Recorder:
    """Stands in for the HIP entry point so that the product glue can be checked on CPU tensors."""

    def __init__(self):
        self.calls = []

    def __call__(self, means3D, scales, rotations, opacities, shs, colors_precomp, viewmatrix, projmatrix, tanfov,
                 bg, H, W, sh_degree, scale_modifier=1.0, enable_cov_grad=True, enable_sh_grad=True, means2D=None,
                 max_pairs=None, view_scale=None):
        self.calls.append(dict(means3D=means3D, scales=scales, rotations=rotations, opacities=opacities, shs=shs,
                               colors_precomp=colors_precomp, viewmatrix=viewmatrix, projmatrix=projmatrix,
                               tanfov=tanfov, bg=bg, H=H, W=W, sh_degree=sh_degree, scale_modifier=scale_modifier,
                               view_scale=view_scale))
        S, V = viewmatrix.shape[:2]
        z = lambda c: torch.zeros(S, V, c, H, W)
        return z(3), z(1) + 2.0, z(1), torch.zeros(S, V, means3D.shape[1], dtype=torch.int32)


def _expand_batched_call(c: dict) -> list[dict]:
    """What the batched call means per (scene, view): the view_scale is applied to means and scales."""
    out = []
    S, V = c["viewmatrix"].shape[:2]
    for s in range(S):
        for v in range(V):
            k = 1.0 if c["view_scale"] is None else c["view_scale"][s, v]
            bg = c["bg"] if c["bg"].dim() == 1 else c["bg"][s, v]
            out.append(dict(
                image_height=c["H"], image_width=c["W"], tanfovx=float(c["tanfov"][s, v, 0]),
                tanfovy=float(c["tanfov"][s, v, 1]), bg=bg, scale_modifier=c["scale_modifier"],
                projmatrix=c["projmatrix"][s, v], sh_degree=c["sh_degree"], means3D=c["means3D"][s] * k,
                shs=None if c["shs"] is None else c["shs"][s],
                colors_precomp=None if c["colors_precomp"] is None else c["colors_precomp"][s],
                opacities=c["opacities"][s].reshape(-1, 1), scales=c["scales"][s] * k, rotations=c["rotations"][s],
                viewmatrix=c["viewmatrix"][s, v]))
    return out


def _expected_camera(g):
    """(view [N,4,4], proj [N,4,4], tanfov [N,2], scale [N]) the reference used, from the recorded calls."""
    i = g["inputs"]
    view = torch.stack([c["kwargs"]["viewmatrix"] for c in g["calls"]])
    proj = torch.stack([c["settings"]["projmatrix"] for c in g["calls"]])
    tanfov = torch.tensor([[c["settings"]["tanfovx"], c["settings"]["tanfovy"]] for c in g["calls"]])
    scale = (1 / i["near"]).reshape(-1) if i["make_scale_invariant"] else torch.ones(i["near"].numel())
    return view, proj, tanfov, scale


@pytest.mark.parametrize("tag", ["decoder_k4_si", "decoder_k25_nosi"])
def test_product_torch_camera_path_matches_reference_callsite(golden_dir, tag):
    """spfsplatv2_amd.camera_tensors (the torch preamble offered to direct users of rasterize_batch)."""
    from spfsplatv2_amd import decoder as dec
    g = torch.load(golden_dir / f"callsite_{tag}.pt")
    i = g["inputs"]
    b, v = i["extrinsics"].shape[:2]
    view, proj, tanfov, scale = dec.camera_tensors(i["extrinsics"].reshape(b * v, 4, 4),
                                                   i["intrinsics"].reshape(b * v, 3, 3), i["near"].reshape(-1),
                                                   i["far"].reshape(-1), i["make_scale_invariant"])
    ev, ep, et, es = _expected_camera(g)
    _close(view, ev); _close(proj, ep); _close(tanfov, et); _close(scale, es)
    # the Gaussians the reference passed are the scene's Gaussians times that scale
    for n, c in enumerate(g["calls"]):
        _close(i["means"][n // v] * scale[n], c["kwargs"]["means3D"])
        _close(i["scales"][n // v] * scale[n], c["kwargs"]["scales"])
        _close(i["harmonics"][n // v].transpose(-1, -2), c["kwargs"]["shs"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["decoder_k4_si", "decoder_k25_nosi"])
def test_hip_camera_kernel_matches_reference_callsite(hip_lib, golden_dir, tag):
    """The fused HIP camera kernel (what DecoderSplattingCUDA.forward runs) against the recorded call site."""
    import spfsplatv2_amd as spf
    g = torch.load(golden_dir / f"callsite_{tag}.pt")
    i = g["inputs"]
    view, proj, tanfov, scale = spf.camera_forward(i["extrinsics"].cuda(), i["intrinsics"].cuda(), i["near"].cuda(),
                                                   i["far"].cuda(), i["make_scale_invariant"])
    ev, ep, et, es = _expected_camera(g)
    _close(view.cpu().reshape(-1, 4, 4), ev, 2e-6); _close(proj.cpu().reshape(-1, 4, 4), ep, 2e-6)
    _close(tanfov.cpu().reshape(-1, 2), et, 2e-6); _close(scale.cpu().reshape(-1), es, 2e-6)


def test_product_orthographic_callsite(golden_dir, monkeypatch):
    from spfsplatv2_amd import decoder as dec
    g = torch.load(golden_dir / "callsite_render_cuda.pt")
    i = g["inputs"]
    rec = _Recorder()
    monkeypatch.setattr(dec, "rasterize_batch", rec)
    out = dec.render_cuda_orthographic(i["extrinsics"], i["ortho_width"], i["ortho_height"], i["near"], i["far"],
                                       i["ortho_image_shape"], torch.zeros(3, 3), i["means"], i["covariances"],
                                       i["harmonics"], i["opacities"], i["rotations"], i["scales"])
    assert out.shape == (3, 3, *i["ortho_image_shape"])
    for a, w in zip(_expand_batched_call(rec.calls[0]), g["calls_ortho"]):
        _check_call(a, w)


def test_product_decoder_registry_and_types():
    from spfsplatv2_amd import decoder as dec
    cfg = dec.DecoderSplattingCUDACfg(name="splatting_cuda", background_color=[0.0, 0.0, 0.0],
                                      make_scale_invariant=True, enable_cov_grad=True, enable_sh_grad=True)
    d = dec.get_decoder(cfg)
    assert isinstance(d, dec.DecoderSplattingCUDA) and isinstance(d, dec.Decoder)
    assert list(dec.DECODERS) == ["splatting_cuda"] and len(list(d.parameters())) == 0
    assert "background_color" not in d.state_dict()          # non-persistent buffer, like the reference
    # planning: the module plans for itself (slack 1.5) until the caller sets a plan; the caller's plan is the caller's
    from spfsplatv2_amd import PairBudget
    assert d.auto_plan == 1.5 and not d.auto_plan_defer and d.max_pairs is None and not d._auto_owned
    d._set_auto(PairBudget(1000, 512, "early"))
    assert d.max_pairs.capacity == 1000 and d._auto_owned
    mine = PairBudget(2000, 1024, "deferred")
    d.max_pairs = mine
    assert d.max_pairs is mine and not d._auto_owned and d._auto_key is None
    d.max_pairs = None
    assert d.max_pairs is None and not d._auto_owned


def test_auto_plan_environment_switches(monkeypatch):
    from spfsplatv2_amd import decoder as dec
    cfg = dec.DecoderSplattingCUDACfg(name="splatting_cuda", background_color=[0.0, 0.0, 0.0],
                                      make_scale_invariant=True, enable_cov_grad=True, enable_sh_grad=True)
    for env, want in (("0", None), ("", None), ("2", 2.0), ("1.25", 1.25)):
        monkeypatch.setenv("SPF_AUTO_PLAN", env)
        assert dec.get_decoder(cfg).auto_plan == want, env
    for bad in ("fast", "0.5", "-1"):                       # a typo names the variable; a slack below 1 is refused
        monkeypatch.setenv("SPF_AUTO_PLAN", bad)
        with pytest.raises(ValueError, match="SPF_AUTO_PLAN"):
            dec.get_decoder(cfg)
    monkeypatch.delenv("SPF_AUTO_PLAN")
    monkeypatch.setenv("SPF_AUTO_PLAN_DEFER", "1")
    assert dec.get_decoder(cfg).auto_plan_defer and dec.get_decoder(cfg).auto_plan == 1.5


def test_switching_auto_plan_off_drops_the_modules_own_plan():
    """`auto_plan = None` on a live decoder returns to EXACT mode -- the plan the module made for itself (and
    the per-shape stash) goes with the switch, a plan the CALLER set stays; copies and pickles of a decoder carry no
    graphs, events or pinned words."""
    import copy
    import pickle
    from spfsplatv2_amd import PairBudget, decoder as dec
    cfg = dec.DecoderSplattingCUDACfg(name="splatting_cuda", background_color=[0.0, 0.0, 0.0],
                                      make_scale_invariant=True, enable_cov_grad=True, enable_sh_grad=True)
    d = dec.get_decoder(cfg)
    d._auto_key = ("shape",)
    d._set_auto(PairBudget(1000, 512, "deferred"))
    d._stash_auto()
    d.auto_plan = None
    assert d.max_pairs is None and not d._auto_owned and not d._auto_plans and d._auto_key is None
    mine = PairBudget(2000, 1024, "deferred")
    d.max_pairs = mine
    d.auto_plan = 0
    assert d.max_pairs is mine and d.auto_plan is None
    with pytest.raises(ValueError):
        d.auto_plan = 0.9
    d.auto_plan = 1.5
    d._graphs["k"] = object()                                # (stand-ins for a captured graph / a pinned word + event)
    d._auto_verdict = (object(), object())
    for clone in (copy.deepcopy(d), pickle.loads(pickle.dumps(d))):
        assert clone._graphs == {} and clone._auto_verdict is None and clone.auto_plan == 1.5
        assert clone.max_pairs == mine and torch.equal(clone.background_color, d.background_color)
    assert "k" in d._graphs


@pytest.mark.parametrize("tag", ["decoder_k4_si", "decoder_k25_nosi"])
def test_decoder_module_postprocessing_matches_reference(golden_dir, monkeypatch, tag):
    """A2: `DecoderSplattingCUDA.forward` itself -- shapes and its `depth x near` post-processing
    (decoder_splatting_cuda.py:66-78) -- against what the reference's own module returned around a stand-in
    rasterizer whose depth is `2 + call number` (tests/golden/make_callsite_goldens.py)."""
    from spfsplatv2_amd import decoder as dec
    g = torch.load(golden_dir / f"callsite_{tag}.pt")
    i = g["inputs"]
    seen = {}

    def fake_render_batch(extrinsics, intrinsics, near, far, means3D, scales, rotations, opacities, shs, colors, bg,
                          H, W, sh_degree, scale_invariant, *a, **kw):
        S, V = extrinsics.shape[:2]
        seen.update(scale_invariant=scale_invariant, sh_degree=sh_degree, shs=shs)
        call = torch.arange(1, S * V + 1, dtype=torch.float32).reshape(S, V, 1, 1, 1)   # the reference's call order
        return (torch.zeros(S, V, 3, H, W), (2.0 + call).expand(S, V, 1, H, W).clone(), torch.zeros(S, V, 1, H, W),
                torch.zeros(S, V, means3D.shape[1], dtype=torch.int32))

    monkeypatch.setattr(dec, "render_batch", fake_render_batch)
    d = dec.get_decoder(dec.DecoderSplattingCUDACfg(name="splatting_cuda", background_color=i["background_color"],
                                                    make_scale_invariant=i["make_scale_invariant"],
                                                    enable_cov_grad=True, enable_sh_grad=True))
    gs = dec.Gaussians(i["means"], i["covariances"], i["rotations"], i["scales"], i["harmonics"], i["opacities"])
    out = d.forward(gs, i["extrinsics"], i["intrinsics"], i["near"], i["far"], i["image_shape"], depth_mode="depth")
    assert isinstance(out, dec.DecoderOutput) and tuple(out.color.shape) == g["decoder_color_shape"]
    assert out.depth.shape == g["decoder_depth"].shape and torch.equal(out.depth, g["decoder_depth"])
    assert seen["scale_invariant"] == i["make_scale_invariant"]
    assert seen["sh_degree"] == g["calls"][0]["settings"]["sh_degree"]
