"""development: HOST cost of ONE per-view drop-in call (GaussianRasterizer(settings)(...) + backward), cProfile top list"""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import spfsplatv2_amd as spf
from spfsplatv2_amd import decoder as dec, synthetic as syn

dev = torch.device("cuda", 0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
b = syn.make_batch("C2", 1, 1, seed=1, G=G).to(dev)
h, w = b.image_shape
view, proj, tanfov, scale = dec.camera_tensors(b.extrinsics[:, 0], b.intrinsics[:, 0], b.near[:, 0], b.far[:, 0])
means = (b.means[0] * scale[0]).requires_grad_(True)
scales = (b.scales[0] * scale[0]).requires_grad_(True)
rot = b.rotations[0].clone().requires_grad_(True)
opac = b.opacities[0, :, None].clone().requires_grad_(True)
shs = b.harmonics[0].transpose(-1, -2).contiguous().requires_grad_(True)
vm = view[0].clone().requires_grad_(True)
bg = torch.zeros(3, device=dev)
tx, ty = float(tanfov[0, 0]), float(tanfov[0, 1])
target = b.target[0, 0]


def call():
    for t in (means, scales, rot, opac, shs, vm):
        t.grad = None
    m2d = torch.zeros_like(means, requires_grad=True)
    settings = spf.GaussianRasterizationSettings(image_height=h, image_width=w, tanfovx=tx, tanfovy=ty, bg=bg,
                                                 scale_modifier=1.0, projmatrix=proj[0], sh_degree=0, prefiltered=False,
                                                 debug=False, enable_cov_grad=True, enable_sh_grad=True)
    image = spf.GaussianRasterizer(settings)(means3D=means, means2D=m2d, shs=shs, colors_precomp=None, opacities=opac,
                                             scales=scales, rotations=rot, viewmatrix=vm)[0]
    ((image - target) ** 2).mean().backward()


for _ in range(20):
    call()
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    call()
torch.cuda.synchronize()
print(f"per call (fwd + loss + bwd, exact mode): {1e3 * (time.perf_counter() - t0) / N:.4f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    call()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
