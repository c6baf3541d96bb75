"""Turn a real `diff_gauss_pose` installation into golden vectors that pin oracle/splat_ref.py (oracle/PINNING.md).

TEST INFRASTRUCTURE; cannot run in the build container (the package is not available offline).  Run it where the
reference's environment exists:

    python oracle/make_pinning_goldens.py [--device cuda] [--out tests/golden/diff_gauss_pose_pins.pt]

For a handful of seeded synthetic scenes (spfsplatv2_amd/synthetic.py -- CPU tensors, no product code on the path) it
calls the package exactly as the reference does (/root/reference/src/model/decoder/cuda_splatting.py:105-138) and stores
inputs, all six outputs, and the gradient of a fixed scalar loss w.r.t. every differentiable input.  The cases are built
so that each convention of PINNING.md's table is observable in at least one of them.
"""
from __future__ import annotations

import argparse
import subprocess
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

GRAD_INPUTS = ("means3D", "shs", "colors_precomp", "opacities", "scales", "rotations", "viewmatrix")


def cases():
    from oracle import glue_ref
    from spfsplatv2_amd import synthetic as syn
    out = {}

    def add(name, batch, bg, scale_invariant=True, use_sh=True, mutate=None, zoom=1.0):
        intr = batch.intrinsics.clone()
        intr[..., 0, 0] *= zoom
        intr[..., 1, 1] *= zoom
        args = glue_ref.callsite_args(batch.extrinsics[:, 0], intr[:, 0], batch.near[:, 0], batch.far[:, 0],
                                      batch.image_shape, torch.tensor([bg]), batch.means, batch.harmonics,
                                      batch.opacities, batch.rotations, batch.scales, scale_invariant, use_sh)[0]
        if mutate:
            mutate(args)
        out[name] = dict(args=args, target=batch.target[0, 0])

    mk = syn.make_batch
    add("plain_k1", mk("C1", 1, 1, seed=1, s_mult=30.0), (0.0, 0.0, 0.0))                       # B#2,5,6,10,11
    add("bg_depth_alpha_k4", mk("TEST", 1, 1, seed=3, s_mult=8.0, G=1500, K=4, image_hw=(80, 112)), (0.1, 0.2, 0.3))
    add("sh3_k16", mk("TEST", 1, 1, seed=5, s_mult=20.0, G=2048, K=16, image_hw=(64, 64)), (1.0, 1.0, 1.0))   # B#9
    add("sh4_k25", mk("TEST", 1, 1, seed=6, s_mult=10.0, G=700, K=25, image_hw=(48, 48)), (0.0, 0.0, 0.0),
        mutate=lambda a: a["shs"][:, 1:].mul_(4.0))                                             # B#9b
    add("sh4_k25_band4_zeroed", mk("TEST", 1, 1, seed=6, s_mult=10.0, G=700, K=25, image_hw=(48, 48)), (0.0, 0.0, 0.0),
        mutate=lambda a: (a["shs"][:, 1:].mul_(4.0), a["shs"][:, 16:].zero_()))
    add("zoom_jacobian_clamp", mk("C1", 1, 1, seed=12, s_mult=120.0, K=4), (0.2, 0.4, 0.6), zoom=2.2)   # B#4
    add("opaque_alpha_clamp", mk("C1", 1, 1, seed=13, s_mult=60.0),(0.0, 0.0, 0.0),
        mutate=lambda a: a["opacities"].fill_(0.9995))                                          # B#10
    add("dark_colour_clamp", mk("C1", 1, 1, seed=14, s_mult=60.0, K=4), (0.0, 0.0, 0.0),
        mutate=lambda a: a["shs"][:, 0].sub_(1.7))                                              # B#9
    add("raw_quaternions", mk("C1", 1, 1, seed=15, s_mult=60.0), (0.0, 0.0, 0.0),
        mutate=lambda a: a["rotations"].mul_(torch.linspace(0.5, 1.5, a["rotations"].shape[0])[:, None]))   # B#7
    add("near_cull", mk("C1", 1, 1, seed=16, s_mult=30.0), (0.0, 0.0, 0.0),
        mutate=lambda a: a["means3D"][:64, 2].copy_(torch.linspace(0.15, 0.25, 64)))            # B#3
    add("colors_precomp", mk("TEST", 1, 1, seed=31, s_mult=12.0, G=900, K=1, image_hw=(48, 80)), (0.3, 0.1, 0.2),
        use_sh=False, mutate=lambda a: a["colors_precomp"].abs_())
    add("many_layers_T_stop", mk("TESTBIG", 1, 1, seed=21, G=12000), (0.0, 0.0, 0.0))          # B#10 (T < 1e-4 stop)
    return out


def run_case(dg, dev, c):
    a, target = c["args"], c["target"].to(dev)
    leaves = {k: (a[k].to(dev).clone().requires_grad_(True) if a.get(k) is not None else None) for k in GRAD_INPUTS}
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    settings = dg.GaussianRasterizationSettings(
        image_height=a["image_height"], image_width=a["image_width"], tanfovx=a["tanfovx"], tanfovy=a["tanfovy"],
        bg=a["bg"].to(dev), scale_modifier=1.0, projmatrix=a["projmatrix"].to(dev), sh_degree=a["sh_degree"],
        prefiltered=False, debug=False, enable_cov_grad=True, enable_sh_grad=True)
    image, depth, norm, alpha, radii, extra = dg.GaussianRasterizer(settings)(
        means3D=leaves["means3D"], means2D=means2D, shs=leaves["shs"], colors_precomp=leaves["colors_precomp"],
        opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"],
        viewmatrix=leaves["viewmatrix"])
    gen = torch.Generator().manual_seed(99)
    wd = torch.rand(depth.shape, generator=gen).to(dev)
    loss = ((image - target) ** 2).mean() + 0.01 * (depth * wd).mean()
    loss.backward()
    cpu = lambda t: None if t is None else t.detach().cpu()
    return dict(inputs={k: cpu(v) for k, v in a.items() if isinstance(v, torch.Tensor)},
                settings={k: a[k] for k in ("image_height", "image_width", "tanfovx", "tanfovy", "sh_degree")},
                target=c["target"], depth_weight=wd.cpu(), image=cpu(image), depth=cpu(depth), alpha=cpu(alpha),
                radii=cpu(radii), norm_is_none=norm is None, loss=float(loss),
                grads={k: cpu(v.grad) for k, v in leaves.items() if v is not None}, means2D_grad=cpu(means2D.grad))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--out", default=str(ROOT / "tests" / "golden" / "diff_gauss_pose_pins.pt"))
    args = ap.parse_args()
    import diff_gauss_pose as dg          # the real package: requirements.txt:88
    try:
        pkg_dir = Path(dg.__file__).resolve().parent
        rev = subprocess.run(["git", "-C", str(pkg_dir), "rev-parse", "HEAD"], capture_output=True, text=True).stdout
    except Exception:
        rev = ""
    out = {"package": getattr(dg, "__file__", "?"), "revision": rev.strip(), "cases": {}}
    for name, c in cases().items():
        out["cases"][name] = run_case(dg, args.device, c)
        print(name, "loss", out["cases"][name]["loss"])
    torch.save(out, args.out)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
