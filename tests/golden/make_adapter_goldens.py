"""Golden vectors of the reference's UnifiedGaussianAdapter.forward (build container only).
Its module imports `jaxtyping` and `src.misc.sh_rotation` (e3nn) which are not installed; both are only names at
import time for this code path, so they are stubbed.       python tests/golden/make_adapter_goldens.py"""
import importlib.util
import sys
import types
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference/src")
jt = types.ModuleType("jaxtyping")
class _Ann:
    def __class_getitem__(cls, item): return cls
for n in ("Float", "Int64", "Bool", "Shaped", "Int"): setattr(jt, n, type(n, (_Ann,), {}))
sys.modules["jaxtyping"] = jt
def pkg(name):
    m = types.ModuleType(name); m.__path__ = []; sys.modules[name] = m; return m
for p in ("src", "src.geometry", "src.misc", "src.model", "src.model.encoder", "src.model.encoder.common"): pkg(p)
sh = types.ModuleType("src.misc.sh_rotation"); sh.rotate_sh = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("unused"))
sys.modules["src.misc.sh_rotation"] = sh
def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, REF / rel); m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m; spec.loader.exec_module(m); return m
load("src.geometry.projection", "geometry/projection.py")
load("src.model.encoder.common.gaussians", "model/encoder/common/gaussians.py")
ga = load("src.model.encoder.common.gaussian_adapter", "model/encoder/common/gaussian_adapter.py")

gen = torch.Generator().manual_seed(8)
out = {}
for deg in (0, 2, 4):
    ad = ga.UnifiedGaussianAdapter(ga.GaussianAdapterCfg(gaussian_scale_min=0.5, gaussian_scale_max=15.0, sh_degree=deg))
    raw = (torch.randn(2, 37, ad.d_in, generator=gen) * 3.0).requires_grad_(True)
    raw.data[0, 0, :3] = torch.tensor([25.0, -30.0, 6.0])        # softplus linear branch / tiny / clamp at 0.3
    raw.data[0, 1, :3] = 400.0
    means = torch.randn(2, 37, 3, generator=gen); opac = torch.rand(2, 37, generator=gen)
    g = ad.forward(means, opac, raw)
    w = [torch.randn(t.shape, generator=gen) for t in (g.scales, g.rotations, g.harmonics)]
    ((g.scales * w[0]).sum() + (g.rotations * w[1]).sum() + (g.harmonics * w[2]).sum()).backward()
    out[f"deg{deg}"] = {"raw": raw.detach().clone(), "means": means, "opacities": opac, "scales": g.scales.detach(),
                        "rotations": g.rotations.detach(), "harmonics": g.harmonics.detach(),
                        "covariances": g.covariances.detach(), "sh_mask": ad.sh_mask.clone(), "weights": w,
                        "raw_grad": raw.grad.clone()}
torch.save(out, HERE / "adapter_goldens.pt")
print("wrote adapter_goldens.pt", {k: tuple(v["raw"].shape) for k, v in out.items()})
