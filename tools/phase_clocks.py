"""Per-phase shader-clock breakdown of spf_render_bwd_lists_kernel (profiling build only):

    SPF_HIPCC_EXTRA=-DSPF_PHASE_CLOCKS python -m spfsplatv2_amd.build --force
    python tools/phase_clocks.py [--config C2] [--scenes 8] [--views 4]

Prints, per phase, the share of wave-cycles (waiting at the closing barrier is charged to the phase before it).
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import spfsplatv2_amd as spf                    # noqa: E402
from spfsplatv2_amd import _lib, synthetic as syn  # noqa: E402

NAMES = ["prologue (lane sort, loads, zero tail)", "staging (records, boxes, scan, pool clear)", "phase A (scatter)",
         "phase B (pixel replay)", "barrier after B", "phase C (entry sums, record write)"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--scenes", type=int, default=8)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--s-mult", type=float, default=1.0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    b = syn.make_batch(a.config, a.scenes, a.views, seed=1000, s_mult=a.s_mult).to(dev)
    names = ("means", "scales", "rotations", "opacities", "harmonics", "extrinsics")
    leaves = {n: getattr(b, n).clone().requires_grad_(True) for n in names}
    bg = torch.zeros(3, device=dev)
    lib = _lib.load()
    buf = (C.c_ulonglong * 8)()
    for it in range(3):
        if it == 2:
            lib.spf_debug_phase_cycles(buf, 1)
            _lib.stage_timing_enable(True)
        color, _, _ = spf.render_views(leaves["extrinsics"], b.intrinsics, b.near, b.far, b.image_shape, bg,
                                       leaves["means"], leaves["harmonics"], leaves["opacities"], leaves["rotations"],
                                       leaves["scales"], scale_invariant=True)
        torch.nn.functional.mse_loss(color, b.target).backward()
    lib.spf_debug_phase_cycles(buf, 0)
    tot = sum(buf[:6])
    for i, nme in enumerate(NAMES):
        print(f"{nme:48s} {buf[i] / 1e6:10.1f} Mcycles  {100.0 * buf[i] / max(tot, 1):5.1f} %")
    waves = max(buf[7], 1)
    print(f"waves {waves}, mean wave lifetime {buf[6] / waves / 100.0:.2f} us (100 MHz wall clock), "
          f"{tot / waves:.0f} shader cycles -> {tot / max(buf[6], 1) * 100.0:.0f} MHz")
    st = _lib.stage_times()
    print({k: round(v[0] / max(v[1], 1), 4) for k, v in st.items() if v[1]})


if __name__ == "__main__":
    main()
