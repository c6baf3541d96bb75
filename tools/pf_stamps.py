"""development: where a block of the forward projection spends its wall-clock time (profiling build only):

    SPF_LIB_DIR=_C_clk SPF_HIPCC_EXTRA=-DSPF_PHASE_CLOCKS python -m spfsplatv2_amd.build
    SPF_LIB_DIR=_C_clk python tools/pf_stamps.py [config] [scenes] [views]

Stamps of every block's wave 0 (100 MHz): start | view loop done | barrier | reservations back | barrier | keys issued.
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
bg = torch.zeros(3, device=dev)
rec = spf.CallRecord()
plan = None


def step():
    with torch.no_grad():
        spf.render_views(b.extrinsics, b.intrinsics, b.near, b.far, b.image_shape, bg, b.means, b.harmonics, b.opacities,
                         b.rotations, b.scales, scale_invariant=True, record=rec, max_pairs=plan)


step()
torch.cuda.synchronize()
plan = spf.plan_pair_budget(rec, slack=1.25, check="deferred")
for _ in range(5):
    step()
torch.cuda.synchronize()
G = b.means.shape[1]
nblk = ((G + 255) // 256) * S
lib = _lib.load()
buf = (C.c_ulonglong * (8 * nblk))()
assert lib.spf_debug_pf_stamps(buf, nblk) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nblk, 8)[:, :6].astype(np.int64)
t0 = t[:, 0].min()
names = ["view loop", "barrier 1", "reservations", "barrier 2", "keys + pair_off"]
d = np.diff(t, axis=1) / 100.0
print(f"{nblk} blocks; kernel span {(t[:, 5].max() - t0) / 100.0:.1f} us; block lifetime mean {(t[:, 5] - t[:, 0]).mean() / 100.0:.2f} us")
for i, n in enumerate(names):
    print(f"  {n:18s} mean {d[:, i].mean():6.2f} us   p50 {np.median(d[:, i]):6.2f}   p90 {np.percentile(d[:, i], 90):6.2f}   max {d[:, i].max():6.2f}")
st = (t[:, 0] - t0) / 100.0
en = (t[:, 5] - t0) / 100.0
print("  block starts (us): p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(st, [10, 50, 90, 100])))
print("  block ends   (us): p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(en, [10, 50, 90, 100])))
# how many blocks are in flight over time (1 us bins)
span = int(en.max()) + 1
fl = [(int(((st <= x) & (en > x)).sum())) for x in range(0, span, max(1, span // 24))]
print("  blocks in flight every", max(1, span // 24), "us:", fl)
