"""development: does running two half-batches on two streams beat one full batch?  (complementary kernels overlapping)"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import spfsplatv2_amd as spf
from spfsplatv2_amd import synthetic as syn

dev = torch.device("cuda:0")
NAMES = ("means", "scales", "rotations", "opacities", "harmonics", "extrinsics")


class Job:
    def __init__(self, S, V, seed):
        self.b = syn.make_batch("C2", S, V, seed=seed).to(dev)
        self.leaves = {n: getattr(self.b, n).clone().requires_grad_(True) for n in NAMES}
        self.bg = torch.zeros(3, device=dev)
        self.one = torch.ones((), device=dev)
        self.rec = spf.CallRecord()
        self.plan = None

    def step(self):
        b, L = self.b, self.leaves
        for t in L.values():
            t.grad = None
        color, depth, alpha, _ = spf.render_batch(L["extrinsics"], b.intrinsics, b.near, b.far, L["means"], L["scales"],
                                                  L["rotations"], L["opacities"], L["harmonics"], None, self.bg,
                                                  b.image_shape[0], b.image_shape[1], 0, True, True, True,
                                                  max_pairs=self.plan, sh_layout="g3k", record=self.rec)
        spf.mse_loss(color, b.target).backward(gradient=self.one)

    def make_plan(self):
        self.step()
        torch.cuda.synchronize()
        self.plan = spf.plan_pair_budget(self.rec, check="deferred")


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


full = Job(8, 4, 1000); full.make_plan()
a = Job(4, 4, 2000); a.make_plan()
b = Job(4, 4, 3000); b.make_plan()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def seq():
    a.step(); b.step()


def par():
    with torch.cuda.stream(s1):
        a.step()
    with torch.cuda.stream(s2):
        b.step()


print("full 8x4 step        %.4f ms" % timeit(full.step))
print("two 4x4 sequential   %.4f ms" % timeit(seq))
print("two 4x4 on 2 streams %.4f ms" % timeit(par))
q = [Job(2, 4, 4000 + i) for i in range(4)]
for j in q:
    j.make_plan()
ss = [torch.cuda.Stream() for _ in range(4)]


def par4():
    for j, s in zip(q, ss):
        with torch.cuda.stream(s):
            j.step()


print("four 2x4 on 4 streams %.4f ms" % timeit(par4))

# ---- the same with HIP graphs (no host launch cost) ----
def capture(job, stream):
    for t in job.leaves.values():
        t.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        job.step()
    return g


torch.cuda.synchronize()
gf = capture(full, torch.cuda.Stream())
ga = capture(a, s1)
gb = capture(b, s2)
torch.cuda.synchronize()


def gseq():
    ga.replay(); gb.replay()


def gpar():
    with torch.cuda.stream(s1):
        ga.replay()
    with torch.cuda.stream(s2):
        gb.replay()


print("graph: full 8x4            %.4f ms" % timeit(gf.replay))
print("graph: two 4x4 sequential  %.4f ms" % timeit(gseq))
print("graph: two 4x4, 2 streams  %.4f ms" % timeit(gpar))
