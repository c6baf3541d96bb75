import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import spfsplatv2_amd as spf
from spfsplatv2_amd import synthetic as syn
dev = torch.device("cuda", 0)
b = syn.make_batch("C2", 8, 4, seed=1000).to(dev)
h, w = b.image_shape
names = ("means", "scales", "rotations", "opacities", "harmonics", "extrinsics")
leaves = {n: getattr(b, n).clone().requires_grad_(True) for n in names}
bg = torch.zeros(3, device=dev)
def step(mp):
    for t in leaves.values(): t.grad = None
    color, depth, _ = spf.render_views(leaves["extrinsics"], b.intrinsics, b.near, b.far, (h, w), bg, leaves["means"], leaves["harmonics"], leaves["opacities"], leaves["rotations"], leaves["scales"], scale_invariant=True, enable_cov_grad=True, enable_sh_grad=True, max_pairs=mp)
    loss = torch.nn.functional.mse_loss(color, b.target)
    loss.backward()
step(None); torch.cuda.synchronize()
D = spf.last_forward_stats()["num_pairs"]; mp = int(D*1.25)
for _ in range(5): step(mp)
torch.cuda.synchronize()
N=50
t0=time.perf_counter()
for _ in range(N): step(mp)
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print(f"sync-free: host enqueue {1e3*(t1-t0)/N:.3f} ms/step, total {1e3*(t2-t0)/N:.3f} ms/step")
t0=time.perf_counter()
for _ in range(N): step(None)
torch.cuda.synchronize()
t2=time.perf_counter()
print(f"exact: total {1e3*(t2-t0)/N:.3f} ms/step")
# forward only host cost breakdown
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for _ in range(20): step(mp)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
