"""Shared helpers for the parity tests (tests are the only place that may use oracle/)."""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]


def rope_oracle_lib():
    so = ROOT / "oracle" / "_build" / "librope_ref.so"
    if not so.exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    lib = ctypes.CDLL(str(so))
    lib.rope2d_ref_f32.restype = None
    lib.rope2d_ref_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_float]
    return lib


def rope_oracle(tokens_bnhd: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> torch.Tensor:
    """Out-of-place call of the C oracle on a contiguous float32 copy of tokens [B,N,H,D]."""
    lib = rope_oracle_lib()
    t = tokens_bnhd.detach().to(torch.float32).contiguous().clone()
    p = positions.detach().to(torch.int64).contiguous()
    B, N, H, D = t.shape
    lib.rope2d_ref_f32(t.data_ptr(), p.data_ptr(), B, N, H, D, t.stride(0), t.stride(1), base, fwd)
    return t


def rel_linf(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| / max(max|b|, tiny): error relative to the tensor's scale."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def rel_elementwise(a: torch.Tensor, b: torch.Tensor, floor: float = 1e-2) -> float:
    """max over the entries with |b| >= floor * max|b| of |a-b| / |b|: the per-ELEMENT relative error of every entry
    that is not two orders below the tensor's largest (rel_linf alone lets such an entry be 10 % off unseen)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    big = b.abs() >= floor * max(float(b.abs().max()), 1e-300)
    if not bool(big.any()):
        return 0.0
    return float(((a - b).abs()[big] / b.abs()[big]).max())


# ---------------------------------------------------------------------------------------------
# Rasterizer parity plumbing
# ---------------------------------------------------------------------------------------------
GRAD_NAMES = ("means", "scales", "rotations", "opacities", "harmonics", "extrinsics")


def loss_weights(batch, seed=1234):
    """Fixed random weights so that depth and alpha outputs also receive non-trivial gradients."""
    gen = torch.Generator().manual_seed(seed)
    b, v = batch.extrinsics.shape[:2]
    h, w = batch.image_shape
    return torch.rand(b, v, h, w, generator=gen), torch.rand(b, v, h, w, generator=gen)


def scalar_loss(color, depth_bvhw, alpha_bv1hw, target, wd, wa, mask=None):
    """MSE on colour (the reference's loss, loss_mse.py:48-51) + weighted depth and alpha terms.  `mask` [b,v,h,w]
    (0/1) switches pixels off: used to keep knife-edge pixels, where a branch of the algorithm may legitimately flip
    between float32 and float64, out of the GRADIENT comparison too (their dL/dpixel is then 0 on both sides)."""
    if mask is None:
        return ((color - target) ** 2).mean() + 0.01 * (depth_bvhw * wd).mean() + \
            0.1 * (alpha_bv1hw[:, :, 0] * wa).mean()
    m = mask.to(color.dtype)
    return (((color - target) ** 2) * m[:, :, None]).mean() + 0.01 * (depth_bvhw * wd * m).mean() + \
        0.1 * (alpha_bv1hw[:, :, 0] * wa * m).mean()


def run_oracle(batch, dtype=torch.float64, background=(0.0, 0.0, 0.0), scale_invariant=True, want_fragile=True,
               with_grads=True, mask_fragile=False, band4=False, grad_names=GRAD_NAMES, pixel_mask=None,
               unmasked_too=False):
    """`mask_fragile`: the loss ignores the pixels the oracle flags as knife-edge; the mask comes back as
    res["pixel_mask"] for `run_product(..., pixel_mask=...)`.  `grad_names`: the inputs that require grad (the others
    are constants, e.g. only "extrinsics" for the reference's test-time pose alignment).  `pixel_mask`: use THIS mask
    in the loss (a float32 evaluation of the oracle held against the float64 one's mask: `float32_resolvable`).
    `unmasked_too`: also res["grads_all"] / res["loss_all"], the gradients of the loss over ALL pixels (same forward)."""
    from oracle import glue_ref
    leaves = {n: getattr(batch, n).detach().clone().to(dtype).requires_grad_(with_grads and n in grad_names)
              for n in GRAD_NAMES}
    # the rasterizer consumes float32 inputs: keep the float32 values exactly, evaluate in `dtype`
    out = glue_ref.decoder_forward(leaves["means"], leaves["harmonics"], leaves["opacities"], leaves["rotations"],
                                   leaves["scales"], leaves["extrinsics"], batch.intrinsics.to(dtype),
                                   batch.near.to(dtype), batch.far.to(dtype), batch.image_shape, background,
                                   make_scale_invariant=scale_invariant, dtype=dtype, want_fragile=want_fragile,
                                   band4=band4, want_radii_fragile=want_fragile)
    color, depth, alpha, radii = out[:4]
    res = dict(color=color.detach(), depth=depth.detach(), alpha=alpha.detach(), radii=radii,
               fragile=out[4] if want_fragile else None, radii_fragile=out[5] if want_fragile else None)
    if with_grads:
        wd, wa = loss_weights(batch)
        mask = (~out[4]).to(torch.float32) if (mask_fragile and want_fragile) else pixel_mask
        res["pixel_mask"] = mask
        loss = scalar_loss(color, depth, alpha, batch.target.to(dtype), wd.to(dtype), wa.to(dtype), mask)
        take = lambda: {n: (leaves[n].grad.detach().clone() if leaves[n].grad is not None
                            else torch.zeros_like(leaves[n])) for n in grad_names}
        if loss.requires_grad:               # (nothing visible in any view: the loss is a constant)
            loss.backward(retain_graph=unmasked_too)
        res["loss"] = float(loss.detach())
        res["grads"] = take()
        if unmasked_too:
            for n in grad_names:
                leaves[n].grad = None
            loss_all = scalar_loss(color, depth, alpha, batch.target.to(dtype), wd.to(dtype), wa.to(dtype), None)
            if loss_all.requires_grad:
                loss_all.backward()
            res["loss_all"] = float(loss_all.detach())
            res["grads_all"] = take()
    return res


class _SplitLeaf:
    """Two leaves standing in for one [.,3,25] harmonics leaf (run_product(split=True)): `.grad` joins them."""

    def __init__(self, low, high):
        self.low, self.high = low, high

    @property
    def grad(self):
        if self.low.grad is None and self.high.grad is None:
            return None
        z = lambda t: torch.zeros_like(t) if t.grad is None else t.grad
        return torch.cat((z(self.low), z(self.high)), dim=-1)

    @grad.setter
    def grad(self, value):
        assert value is None
        self.low.grad = self.high.grad = None


def product_decoder(background=(0.0, 0.0, 0.0), scale_invariant=True, device="cuda", max_pairs=None, band4=None,
                    auto_plan=None):
    """The product's decoder MODULE under the reference's registry name and config
    (decoder/__init__.py:4-12, decoder_splatting_cuda.py:15-21).  `auto_plan`: the module's own planning is OFF here
    unless asked for (the tests pin exact / planned calls themselves; tests/test_gpu_eval_graphs.py covers the default)."""
    from spfsplatv2_amd import decoder as dec
    d = dec.get_decoder(dec.DecoderSplattingCUDACfg(name="splatting_cuda", background_color=list(background),
                                                    make_scale_invariant=scale_invariant, enable_cov_grad=True,
                                                    enable_sh_grad=True)).to(device)
    d.auto_plan = auto_plan
    d.max_pairs = max_pairs
    d.sh_band4 = band4
    return d


def run_product(batch, device="cuda", background=(0.0, 0.0, 0.0), scale_invariant=True, with_grads=True,
                max_pairs=None, pixel_mask=None, band4=None, grad_names=GRAD_NAMES, unmasked_too=False, split=False):
    """The product, end to end THROUGH ITS DECODER MODULE (`DecoderSplattingCUDA.render` = `forward` + the alpha and
    radii the reference's decoder drops): colour and depth are the module's own outputs, including its depth x near
    post-processing (decoder_splatting_cuda.py:72-76) -- nothing of it is re-implemented here.  `split`: the d_sh = 25
    harmonics go in BAND-SPLIT (two leaves, [.,3,16] and [.,3,9]: `Gaussians.harmonics_band4`); their gradients come back
    joined to the [.,3,25] the oracle produces (a plane without a gradient -- band 4 not evaluated -- as zeros)."""
    from spfsplatv2_amd import decoder as dec
    bd = batch.to(device)
    leaves = {n: getattr(bd, n).detach().clone().requires_grad_(with_grads and n in grad_names) for n in GRAD_NAMES}
    d = product_decoder(background, scale_invariant, device, max_pairs, band4)
    if split:
        want = with_grads and "harmonics" in grad_names
        leaves["harmonics"] = _SplitLeaf(bd.harmonics[..., :16].contiguous().requires_grad_(want),
                                         bd.harmonics[..., 16:].contiguous().requires_grad_(want))
        g = dec.Gaussians(leaves["means"], bd.covariances, leaves["rotations"], leaves["scales"], leaves["harmonics"].low,
                          leaves["opacities"], harmonics_band4=leaves["harmonics"].high)
    else:
        g = dec.Gaussians(leaves["means"], bd.covariances, leaves["rotations"], leaves["scales"], leaves["harmonics"],
                          leaves["opacities"])
    out, alpha, radii = d.render(g, leaves["extrinsics"], bd.intrinsics, bd.near, bd.far, bd.image_shape)
    color, depth = out.color, out.depth
    res = dict(color=color.detach().cpu(), depth=depth.detach().cpu(), alpha=alpha.detach().cpu(),
               radii=radii.cpu(), stats={k: v for k, v in d.last_call.items() if k != "counters"}, decoder=d)
    if with_grads:
        wd, wa = loss_weights(batch)
        loss = scalar_loss(color, depth, alpha, bd.target, wd.to(device), wa.to(device),
                           None if pixel_mask is None else pixel_mask.to(device))
        loss.backward(retain_graph=unmasked_too)
        res["loss"] = float(loss.detach())
        res["grads"] = {n: leaves[n].grad.detach().cpu() for n in grad_names}
        if unmasked_too:          # the loss over ALL pixels, second backward through the same forward
            for n in grad_names:
                leaves[n].grad = None
            loss_all = scalar_loss(color, depth, alpha, bd.target, wd.to(device), wa.to(device), None)
            loss_all.backward()
            res["loss_all"] = float(loss_all.detach())
            res["grads_all"] = {n: leaves[n].grad.detach().cpu() for n in grad_names}
    return res


def compare(prod: dict, ref: dict, rgb_tol=1e-4, grad_tol=1e-3, max_fragile_frac=0.005, grad_el_tol=1e-2) -> dict:
    """Returns a report; raises AssertionError with the report if a gate fails.

    Gates (BASELINE.json north_star): RGB within 1e-4 absolute, gradients within 1e-3 of the tensor's scale (`g_*`:
    that IS north_star's gradient tolerance).  `gel_*` (round 4) is an ADDITION of this suite, not a reading of
    north_star: every gradient ENTRY that is at least 1 % of its tensor's largest within 1e-2 of ITSELF -- ten times
    looser than 1e-3 in relative terms, but per element instead of per tensor (the L-infinity-of-scale gate alone lets
    an entry two orders below the maximum be 10 % off).
    Reported, not gated: `rgb_max_all` (the unmasked image) and, when both sides carry them, `gall_*`: the gradients of
    the loss over ALL pixels, knife-edge pixels included (a branch that legitimately flips there moves them).
    Pixels the float64 oracle flags as knife-edge (an alpha within 5e-5 relative of 1/255 plus what the float32 pixel
    centre can move it by, a transmittance within 0.1% of the 1e-4 stop, a contribution that depends on a tile
    membership decided by a footprint radius within 1e-4 of an integer, two contributors closer in depth than float32
    resolves -- their order is decided by float32 depth bits; oracle/splat_ref.py FRAG_*) are excluded from the RGB gate
    -- there a one-ulp difference legitimately flips a branch -- but must stay a small fraction: <= 0.5 % by default
    (measured <= 0.05 % on the fixed cases), and the number of pixels of the UNMASKED image that are off by more than
    the tolerance may not exceed the number flagged.
    """
    frag = ref["fragile"] if ref.get("fragile") is not None else torch.zeros_like(ref["depth"], dtype=torch.bool)
    ok = ~frag
    rep = {"fragile_frac": float(frag.float().mean())}
    d = (prod["color"].double() - ref["color"].double()).abs()
    rep["rgb_max_all"] = float(d.max())
    rep["rgb_max"] = float((d * ok[:, :, None]).max())
    # the UNMASKED image: pixels off by more than the tolerance (any channel) may not outnumber the pixels the oracle
    # flagged -- the mask cannot hide a real disagreement larger than itself
    rep["bad_frac_all"] = float((d.amax(dim=2) > rgb_tol).float().mean())
    # radii are integer work: bit-exact wherever the oracle does not flag a rounding knife-edge of the Gaussian itself
    if prod.get("radii") is not None and ref.get("radii_fragile") is not None:
        rok = ~ref["radii_fragile"]
        rep["radii_fragile_frac"] = float(ref["radii_fragile"].float().mean())
        rep["radii_mismatch"] = int(((prod["radii"].long() != ref["radii"].long()) & rok).sum())
    dd = (prod["depth"].double() - ref["depth"].double()).abs() * ok
    rep["depth_rel"] = float(dd.max() / max(float(ref["depth"].abs().max()), 1e-30))
    rep["alpha_max"] = float(((prod["alpha"].double() - ref["alpha"].double()).abs() * ok[:, :, None]).max())
    if "grads" in prod and "grads" in ref:
        for n in ref["grads"]:
            rep["g_" + n] = rel_linf(prod["grads"][n], ref["grads"][n])
            rep["gel_" + n] = rel_elementwise(prod["grads"][n], ref["grads"][n])
    if "grads_all" in prod and "grads_all" in ref:
        for n in ref["grads_all"]:
            rep["gall_" + n] = rel_linf(prod["grads_all"][n], ref["grads_all"][n])
    fails = []
    if rep["fragile_frac"] > max_fragile_frac:
        fails.append("fragile_frac")
    if rep["bad_frac_all"] > rep["fragile_frac"]:
        fails.append("bad_frac_all")
    if rep.get("radii_mismatch", 0) > 0:
        fails.append("radii_mismatch")
    if rep.get("radii_fragile_frac", 0.0) > max(0.01, 1.5 / max(ref["radii"].numel(), 1)):   # (one of a handful is no signal)
        fails.append("radii_fragile_frac")
    if rep["rgb_max"] > rgb_tol:
        fails.append("rgb_max")
    if rep["depth_rel"] > 1e-4:
        fails.append("depth_rel")
    if rep["alpha_max"] > rgb_tol:
        fails.append("alpha_max")
    for n in GRAD_NAMES:
        if rep.get("g_" + n, 0.0) > grad_tol:
            fails.append("g_" + n)
        if rep.get("gel_" + n, 0.0) > grad_el_tol:
            fails.append("gel_" + n)
    rep["fails"] = fails
    return rep


def float32_resolvable(batch, ref: dict, **kw) -> dict:
    """Second arbiter, for cases that fail `compare`: the SAME restatement of the published algorithm evaluated in
    float32 on the CPU (oracle/splat_ref.py is dtype-generic), held against the float64 values through the same gates
    with the same pixel mask.  If a plain float32 evaluation of the classic formulas cannot meet the tolerances on an
    input, a float64 oracle cannot decide parity with a float32 reference there: the case is `unresolvable`, and is
    reported as such, with both errors -- it is not counted as agreement and no flag of the oracle is involved."""
    f32 = run_oracle(batch, torch.float32, want_fragile=False, pixel_mask=ref.get("pixel_mask"), **kw)
    f32["radii"] = None
    rep = compare(f32, ref, max_fragile_frac=1.0)
    return rep
