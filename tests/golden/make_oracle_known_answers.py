"""Freeze the rasterizer oracle's OWN values (SURVEY.md 8c fixture (3); BASELINE.md "C1 ... known-answer fixture").

    python tests/golden/make_oracle_known_answers.py                -> tests/golden/oracle_known_answers.pt
    python tests/golden/make_oracle_known_answers.py --masks-only   -> the same file with ONLY the knife-edge masks
                                                                       replaced (every frozen VALUE is first checked
                                                                       against today's oracle to 1e-12 and kept as is)

The rasterizer's arithmetic is parity-unpinned (oracle/splat_ref.py header): no reference binary, no reference-held
vector.  The one integrity property the parity story can have is that the checker does not drift towards the kernels
it checks.  This fixture makes that checkable: for BASELINE config 1 and three small TEST scenes it stores the float32
INPUTS and the oracle's float64 outputs -- colour, depth, alpha, radii, all six gradients, the loss -- together with
the knife-edge masks, as they were when this file was generated.  `tests/test_oracle_known_answers.py` (CPU) demands
that today's oracle reproduces every VALUE to 1e-12 and every MASK exactly, so an oracle edit that changes a value or
widens a flag has to regenerate this file -- visibly, in the history -- and `tests/test_gpu_known_answers.py` holds
the HIP kernels against the frozen values rather than against a live oracle run.

The generator needs nothing but this repository (the reference holds no vectors for this path).
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from spfsplatv2_amd import synthetic as syn  # noqa: E402
from tests import util  # noqa: E402

INPUT_FIELDS = ("means", "scales", "rotations", "opacities", "harmonics", "covariances", "extrinsics", "intrinsics",
                "near", "far", "target")


def _wide(batch, seed):
    """Raw (non-unit) quaternions, opacities of exactly 0 and 1, zero scales, fx != fy, off-centre principal point."""
    w = torch.Generator().manual_seed(seed)
    S, G = batch.opacities.shape
    V = batch.extrinsics.shape[1]
    ru = lambda *shape: torch.rand(*shape, generator=w)
    batch.intrinsics[..., 0, 0] = 0.6 + 1.0 * ru(S, V)
    batch.intrinsics[..., 1, 1] = 0.6 + 1.0 * ru(S, V)
    batch.intrinsics[..., 0, 2] = 0.45 + 0.1 * ru(S, V)
    batch.intrinsics[..., 1, 2] = 0.45 + 0.1 * ru(S, V)
    batch.rotations = batch.rotations * (0.5 + 1.5 * ru(S, G, 1))
    edge = ru(S, G)
    batch.opacities = torch.where(edge < 0.03, torch.zeros_like(batch.opacities), batch.opacities)
    batch.opacities = torch.where(edge > 0.97, torch.ones_like(batch.opacities), batch.opacities)
    batch.scales = torch.where(ru(S, G, 3) < 0.02, torch.zeros_like(batch.scales), batch.scales)
    return batch


def cases():
    """name -> (batch, background, scale_invariant, band4)"""
    out = {}
    # BASELINE.json configs[0]: 256 random Gaussians -> one 64x64 view
    out["c1"] = (syn.make_batch("C1", 1, 1, seed=1, s_mult=30.0), (0.0, 0.0, 0.0), True, False)
    out["test_k4_two_views"] = (syn.make_batch("TEST", 1, 2, seed=21, s_mult=8.0, G=600, K=4, image_hw=(48, 64)),
                                (0.1, 0.2, 0.3), True, False)
    out["test_k25_band4_nosi"] = (syn.make_batch("TEST", 1, 1, seed=22, s_mult=12.0, G=400, K=25, image_hw=(40, 40)),
                                  (0.2, 0.5, 0.9), False, True)
    # thin splats tens to hundreds of pixels long (the regime of the float32 determinant), plus the "wide" edge values
    out["test_k16_long_splats_wide"] = (_wide(syn.make_batch("TEST", 1, 1, seed=23, s_mult=60.0, G=500, K=16,
                                                             image_hw=(48, 48)), 23), (0.0, 0.0, 0.0), True, False)
    return out


def render(batch, bg, si, band4):
    ref = util.run_oracle(batch, torch.float64, background=bg, scale_invariant=si, mask_fragile=True, band4=band4)
    return dict(color=ref["color"], depth=ref["depth"], alpha=ref["alpha"], radii=ref["radii"],
                fragile=ref["fragile"], radii_fragile=ref["radii_fragile"], loss=ref["loss"],
                grads={k: v for k, v in ref["grads"].items()})


def masks_only(dst: Path):
    """An oracle edit that moves a knife-edge WINDOW (not a value): keep every frozen value, replace the masks.  Refuses
    if any value differs from the fixture (that would be a change of the arithmetic, to be regenerated in full)."""
    out = torch.load(dst)
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / max(float(b.double().abs().max()), 1e-300))
    for name, (batch, bg, si, band4) in cases().items():
        want = out[name]["expect"]
        got = render(batch, bg, si, band4)
        for k in ("color", "depth", "alpha"):
            assert rel(got[k], want[k]) < 1e-12, (name, k, "a VALUE changed: regenerate in full, and say why")
        assert torch.equal(got["radii"], want["radii"]), name
        # (loss and gradients are taken over the unflagged pixels: they move WITH the masks and are re-frozen with them)
        old_f, old_r = float(want["fragile"].float().mean()), float(want["radii_fragile"].float().mean())
        for k in ("fragile", "radii_fragile", "loss", "grads"):
            want[k] = got[k]
        print(name, "fragile", old_f, "->", float(got["fragile"].float().mean()), "radii_fragile", old_r, "->",
              float(got["radii_fragile"].float().mean()))
    torch.save(out, dst)
    print("rewrote the masks (and the masked loss / gradients) of", dst)


def main():
    torch.set_num_threads(1)                 # one summation order
    dst = Path(__file__).resolve().parent / "oracle_known_answers.pt"
    if "--masks-only" in sys.argv:
        return masks_only(dst)
    out = {}
    for name, (batch, bg, si, band4) in cases().items():
        res = render(batch, bg, si, band4)
        out[name] = dict(inputs={f: getattr(batch, f) for f in INPUT_FIELDS}, image_shape=tuple(batch.image_shape),
                         background=bg, scale_invariant=si, band4=band4, expect=res)
        print(name, "loss", res["loss"], "fragile", float(res["fragile"].float().mean()),
              "visible", int((res["radii"] > 0).sum()), "of", res["radii"].numel())
    torch.save(out, dst)
    print("wrote", dst, dst.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
