"""development: how many blocks of the lists backward are in flight over a launch (profiling build only):

    SPF_LIB_DIR=_C_clk SPF_HIPCC_EXTRA=-DSPF_PHASE_CLOCKS python -m spfsplatv2_amd.build
    SPF_LIB_DIR=_C_clk python tools/block_stamps.py [config] [scenes] [views]
"""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import spfsplatv2_amd as spf                    # noqa: E402
from spfsplatv2_amd import _lib, synthetic as syn  # noqa: E402

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
for _ in range(4):
    step()
torch.cuda.synchronize()
h, w = b.image_shape
nblk = ((S * V * ((h + 15) // 16) * ((w + 15) // 16) + 7) // 8) * 8
lib = _lib.load()
for name in ("backward lists",):
    buf = (C.c_ulonglong * (2 * nblk))()
    assert lib.spf_debug_block_stamps(buf, nblk) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(nblk, 2).astype(np.int64)
    xcd = np.arange(nblk) & 7                         # (the dispatcher places block b on XCD b % 8)
    keep = t[:, 0] > 0
    t, xcd = t[keep], xcd[keep]
    keep = t[:, 1] > t[:, 1].max() - 100000           # (blocks that leave before their stamps keep an older launch's)
    t, xcd = t[keep], xcd[keep]
    t0 = t[:, 0].min()
    st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
    life = en - st
    span = en.max()
    print(f"{name}: {len(t)} blocks, span {span:.1f} us, block lifetime mean {life.mean():.2f} p50 {np.median(life):.2f} "
          f"p90 {np.percentile(life, 90):.2f} max {life.max():.2f} us; sum of lifetimes / span = {life.sum() / span:.0f} blocks in flight on average")
    step_us = max(1.0, span / 28)
    xs = np.arange(0, span, step_us)
    print(f"  in flight every {step_us:.1f} us:", [int(((st <= x) & (en > x)).sum()) for x in xs])
    print("  last block started at %.1f us; blocks ending in the last 10 us: %d" % (st.max(), int((en > span - 10).sum())))
    print("  per XCD: last block ends at (us)", [round(float(en[xcd == x].max()), 1) for x in range(8)],
          " sum of block lifetimes (ms)", [round(float(life[xcd == x].sum()) / 1e3, 2) for x in range(8)])
