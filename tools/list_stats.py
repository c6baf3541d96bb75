"""development: contributor statistics of the default bench workload (hits per pixel, list positions)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import spfsplatv2_amd as spf
from spfsplatv2_amd import synthetic as syn

dev = torch.device("cuda:0")
b = syn.make_batch("C2", 8, 4, seed=1000).to(dev)
leaves = {n: getattr(b, n).clone().requires_grad_(True) for n in ("means", "scales", "rotations", "opacities", "harmonics", "extrinsics")}
color, _, _ = spf.render_views(leaves["extrinsics"], b.intrinsics, b.near, b.far, b.image_shape, torch.zeros(3, device=dev),
                               leaves["means"], leaves["harmonics"], leaves["opacities"], leaves["rotations"],
                               leaves["scales"], scale_invariant=True)
saved = color.grad_fn.saved_tensors
nc = saved[18].view(-1, 2).long()
last, hits = nc[:, 0].float(), nc[:, 1].float()
print("pixels", nc.shape[0], "mean last", last.mean().item(), "mean hits", hits.mean().item(), "max hits", hits.max().item())
print("pairs", spf.last_forward_stats())
h = hits.view(32, 16, 16, 16, 16).permute(0, 1, 3, 2, 4).reshape(-1, 256)      # per tile
print("per-tile: mean of max hits", h.max(1).values.mean().item(), "mean of mean", h.mean(1).mean().item())
hs = h.sort(1, descending=True).values.view(-1, 4, 64)
print("sorted-wave max hits (mean over tiles) per wave:", hs.max(2).values.mean(0).tolist(), "wave means", hs.mean(2).mean(0).tolist())
