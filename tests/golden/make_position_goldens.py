"""Golden vectors for the producer of the RoPE ``positions`` operand, from the REFERENCE's own classes
(build container only): ``PositionGetter`` (croco/blocks.py:207-219) and the extra-token rule of the masked CroCo
backbone (backbone_masked_croco.py:163-172), the latter re-run here through the same two tensor statements on the
reference's positions.  Writes tests/golden/position_goldens.pt.   python tests/golden/make_position_goldens.py
"""
import importlib.util
import sys
import types
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference/src/model/encoder/backbone/croco")


def load_blocks():
    # blocks.py only needs torch at import time for the class used here; give its optional imports empty homes
    for name in ("timm", "timm.models", "timm.models.layers"):
        sys.modules.setdefault(name, types.ModuleType(name))
    spec = importlib.util.spec_from_file_location("ref_croco_blocks", REF / "blocks.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    blocks = load_blocks()
    getter = blocks.PositionGetter()
    out = {}
    for (b, h, w) in [(1, 1, 1), (2, 3, 5), (3, 16, 16), (1, 18, 14), (2, 16, 16)]:
        pos = getter(b, h, w, torch.device("cpu"))
        one = pos[:, 0:1, :].clone()
        one[:, :, 0] += (pos[:, -1, 0].unsqueeze(-1) + 1)
        with_tok = torch.cat((pos, one), dim=1)
        two = with_tok[:, 0:1, :].clone()
        two[:, :, 0] += (with_tok[:, -1, 0].unsqueeze(-1) + 1)
        out[f"{b}x{h}x{w}"] = {"b": b, "h": h, "w": w, "positions": pos, "plus_one_token": with_tok,
                               "plus_two_tokens": torch.cat((with_tok, two), dim=1)}
    torch.save(out, HERE / "position_goldens.pt")
    print("wrote", HERE / "position_goldens.pt", {k: tuple(v["positions"].shape) for k, v in out.items()})


if __name__ == "__main__":
    main()
