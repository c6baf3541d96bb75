"""development: where the host's time goes in one C2 step through the decoder MODULE (training call on a prepared step):
wall time per phase with a device synchronisation only at the end of the step.   python tools/hostcost_module.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import spfsplatv2_amd as spf
from spfsplatv2_amd import decoder as dec, synthetic as syn

dev = torch.device("cuda", 0)
b = syn.make_batch("C2", 8, 4, seed=1000).to(dev)
names = ("means", "scales", "rotations", "opacities", "harmonics", "extrinsics")
L = {n: getattr(b, n).clone().requires_grad_(True) for n in names}
d = dec.get_decoder(dec.DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True, True, True)).to(dev)
if len(sys.argv) > 1 and sys.argv[1] == "defer":
    d.auto_plan_defer = True
g = dec.Gaussians(L["means"], None, L["rotations"], L["scales"], L["harmonics"], L["opacities"])
h, w = b.image_shape
unit = spf.unit_grad(dev)
T = {"zero": 0.0, "forward": 0.0, "loss": 0.0, "backward": 0.0}


def step(timed):
    t0 = time.perf_counter()
    for t in L.values():
        t.grad = None
    t1 = time.perf_counter()
    out = d.forward(g, L["extrinsics"], b.intrinsics, b.near, b.far, (h, w))
    t2 = time.perf_counter()
    loss = spf.mse_loss(out.color, b.target, 1.0)
    t3 = time.perf_counter()
    loss.backward(gradient=unit)
    t4 = time.perf_counter()
    if timed:
        T["zero"] += t1 - t0; T["forward"] += t2 - t1; T["loss"] += t3 - t2; T["backward"] += t4 - t3


for _ in range(10):
    step(False)
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    step(True)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n
print("prepared steps:", len(d._prepared_steps), "ms/step", round(wall * 1e3, 4), {k: round(v / n * 1e6, 1) for k, v in T.items()}, "us host per phase")

if os.environ.get("SPF_CPROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        step(False)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
