"""Time spf_render_bwd_lists_kernel with parts cut out (profiling build only; results are garbage, times are not):

    SPF_HIPCC_EXTRA=-DSPF_ABLATE python -m spfsplatv2_amd.build --force
    python tools/ablate_bwd.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import spfsplatv2_amd as spf                    # noqa: E402
from spfsplatv2_amd import _lib, synthetic as syn  # noqa: E402

CUTS = {0: "full kernel", 1: "return at entry (launch + tile header)", 2: "return after lane sort + pixel loads",
        3: "return after zeroing the tail records", 4: "rounds: staging only", 5: "rounds: staging + phase A",
        6: "everything but phase C", 7: "everything but phase B",
        8: "FORWARD lists kernel: staging only", 9: "FORWARD lists kernel: staging + phase A",
        10: "FORWARD: staging only, records read coalesced instead of gathered",
        11: "FORWARD: staging only, records always from the same 12 KB",
        12: "FORWARD: staging only (same as 8)"}


def main():
    dev = torch.device("cuda:0")
    b = syn.make_batch("C2", 8, 4, seed=1000).to(dev)
    names = ("means", "scales", "rotations", "opacities", "harmonics", "extrinsics")
    leaves = {n: getattr(b, n).clone().requires_grad_(True) for n in names}
    bg = torch.zeros(3, device=dev)
    lib = _lib.load()

    def step():
        color, _, _ = spf.render_views(leaves["extrinsics"], b.intrinsics, b.near, b.far, b.image_shape, bg,
                                       leaves["means"], leaves["harmonics"], leaves["opacities"], leaves["rotations"],
                                       leaves["scales"], scale_invariant=True)
        torch.nn.functional.mse_loss(color, b.target).backward()

    for cut, what in CUTS.items():
        lib.spf_debug_set_ablate(cut)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        _lib.stage_timing_enable(["render_bwd", "render_fwd"])
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        st, sf = _lib.stage_times()["render_bwd"], _lib.stage_times()["render_fwd"]
        _lib.stage_timing_enable(False)
        print(f"cut {cut}: bwd {st[0] / st[1] * 1e3:8.1f} us  fwd {sf[0] / sf[1] * 1e3:8.1f} us   {what}")


if __name__ == "__main__":
    main()
