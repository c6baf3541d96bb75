"""development: per-stage HIP-event times of the default bench step (planned from the first call like bench.py, so the
direct bins are used; results may be garbage in ablation builds -- the plan's flags are not looked at)
    SPF_LIB_DIR=_C_xyz python tools/stage_times.py [config] [scenes] [views]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import spfsplatv2_amd as spf
from spfsplatv2_amd import _lib, synthetic as syn

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
V = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
b = syn.make_batch(cfg, S, V, seed=1000).to(dev)
leaves = {n: getattr(b, n).clone().requires_grad_(True) for n in ("means", "scales", "rotations", "opacities", "harmonics", "extrinsics")}
bg = torch.zeros(3, device=dev)
rec = spf.CallRecord()
plan = None


def step():
    for t in leaves.values():
        t.grad = None
    color, _, _ = spf.render_views(leaves["extrinsics"], b.intrinsics, b.near, b.far, b.image_shape, bg, leaves["means"],
                                   leaves["harmonics"], leaves["opacities"], leaves["rotations"], leaves["scales"],
                                   scale_invariant=True, enable_cov_grad=True, enable_sh_grad=True, record=rec,
                                   max_pairs=plan)
    spf.mse_loss(color, b.target).backward()


step()
torch.cuda.synchronize()
plan = spf.plan_pair_budget(rec, slack=1.25, check="deferred")
for _ in range(3):
    step()
torch.cuda.synchronize()
_lib.stage_timing_enable(True)
for _ in range(20):
    step()
torch.cuda.synchronize()
st = _lib.stage_times()
print(" ".join(f"{k}={v[0] / v[1] * 1e3:.1f}" for k, v in st.items() if v[1]))
