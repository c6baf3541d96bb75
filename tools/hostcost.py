"""development: HOST cost of one step (tiny workload, so the GPU never back-pressures): total, and a cProfile top list"""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import spfsplatv2_amd as spf
from spfsplatv2_amd import synthetic as syn

dev = torch.device("cuda", 0)
b = syn.make_batch("TEST", 8, 4, seed=1, G=512, image_hw=(32, 32)).to(dev)
names = ("means", "scales", "rotations", "opacities", "harmonics", "extrinsics")
leaves = {n: getattr(b, n).clone().requires_grad_(True) for n in names}
bg = torch.zeros(3, device=dev)
one = torch.ones((), device=dev)
rec = spf.CallRecord()
plan = None


def step():
    for t in leaves.values():
        t.grad = None
    color, depth, _ = spf.render_views(leaves["extrinsics"], b.intrinsics, b.near, b.far, b.image_shape, bg,
                                       leaves["means"], leaves["harmonics"], leaves["opacities"], leaves["rotations"],
                                       leaves["scales"], scale_invariant=True, enable_cov_grad=True, enable_sh_grad=True,
                                       max_pairs=plan, record=rec)
    spf.mse_loss(color, b.target).backward(gradient=one)


step(); torch.cuda.synchronize()
plan = spf.plan_pair_budget(rec, check="deferred")
for _ in range(50):
    step()
torch.cuda.synchronize()
N = 500
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host per step: {1e3 * (t1 - t0) / N:.4f} ms (after sync {1e3 * (time.perf_counter() - t0) / N:.4f})")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
