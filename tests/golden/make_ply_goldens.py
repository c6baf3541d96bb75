"""Golden vertex table of the reference's export_ply (build container only; needs /root/reference + scipy).

The reference writes through the `plyfile` package, which is not installed here; a recording stand-in captures the
structured array handed to `PlyElement.describe` (that array IS the file content).   python tests/golden/make_ply_goldens.py
"""
import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parents[1]))
captured = {}
jt = types.ModuleType("jaxtyping")
class _Ann:
    def __class_getitem__(cls, item): return cls
jt.Float = type("Float", (_Ann,), {})
sys.modules["jaxtyping"] = jt
pf = types.ModuleType("plyfile")
class PlyElement:
    @staticmethod
    def describe(elements, name):
        captured["elements"] = elements.copy(); captured["name"] = name
        return elements
class PlyData:
    def __init__(self, els): pass
    def write(self, path): captured["path"] = str(path)
pf.PlyElement, pf.PlyData = PlyElement, PlyData
sys.modules["plyfile"] = pf
spec = importlib.util.spec_from_file_location("ref_ply_export", "/root/reference/src/model/ply_export.py")
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)

from spfsplatv2_amd import synthetic as syn
b = syn.make_batch("TEST", 1, 1, seed=41, s_mult=5.0, G=300, K=4)
gen = torch.Generator().manual_seed(2)
ext = syn.small_pose(gen, 0.7, 30.0)
m.export_ply(ext, b.means[0], b.scales[0], b.rotations[0], b.harmonics[0], b.opacities[0], Path("/tmp/ref_golden.ply"))
el = captured["elements"]
table = np.stack([el[n] for n in el.dtype.names], axis=1).astype(np.float32)
torch.save({"extrinsics": ext, "means": b.means[0], "scales": b.scales[0], "rotations": b.rotations[0],
            "harmonics": b.harmonics[0], "opacities": b.opacities[0], "names": list(el.dtype.names),
            "table": torch.from_numpy(table)}, HERE / "ply_golden.pt")
print("wrote ply_golden.pt", table.shape, el.dtype.names)
