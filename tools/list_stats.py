"""development: tile-list statistics of a bench workload (list lengths per tile, hits per pixel -- the latter only with a
library built with SPF_HIPCC_EXTRA=-DSPF_LANESORT=1 or -DSPF_PHASE_CLOCKS: the default build does not count them).

    python tools/list_stats.py [--config C2] [--scenes 8] [--views 4]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import spfsplatv2_amd as spf                       # noqa: E402
from spfsplatv2_amd import synthetic as syn        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2")
ap.add_argument("--scenes", type=int, default=8)
ap.add_argument("--views", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda:0")
b = syn.make_batch(a.config, a.scenes, a.views, seed=1000).to(dev)
leaves = {n: getattr(b, n).clone().requires_grad_(True)
          for n in ("means", "scales", "rotations", "opacities", "harmonics", "extrinsics")}
color, _, _ = spf.render_views(leaves["extrinsics"], b.intrinsics, b.near, b.far, b.image_shape,
                               torch.zeros(3, device=dev), leaves["means"], leaves["harmonics"], leaves["opacities"],
                               leaves["rotations"], leaves["scales"], scale_invariant=True)
st = spf.last_forward_stats()
print("pairs", st)
RT = st["tiles"]
tiles = next(t for t in color.grad_fn.saved_tensors if t is not None and t.dtype == torch.int32 and t.numel() == 4 * RT + 8)
start = tiles[2 * RT:3 * RT + 1].long().cpu()
n = start[1:] - start[:-1]
print("tile lists: mean %.1f  min %d  max %d" % (n.float().mean(), n.min(), n.max()))
edges = [0, 64, 128, 192, 256, 320, 384, 448, 512, 640, 768, 1024, 2048, 4096, 1 << 30]
for lo, hi in zip(edges[:-1], edges[1:]):
    c = int(((n > lo) & (n <= hi)).sum())
    if c:
        print(f"  {lo + 1:5d} .. {hi if hi < 1 << 30 else 'inf':>5}: {c:6d} tiles ({100.0 * c / n.numel():5.1f} %)")
P = b.image_shape[0] * b.image_shape[1]
nc = next(t for t in color.grad_fn.saved_tensors if t is not None and t.dtype == torch.int32 and t.numel() == 2 * (RT // ((b.image_shape[0] + 15) // 16 * ((b.image_shape[1] + 15) // 16))) * P)
hits = nc.view(-1, 2)[:, 1].float()
print("hits per pixel: mean %.2f max %d" % (hits.mean(), hits.max()))
