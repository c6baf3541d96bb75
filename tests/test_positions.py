"""PositionGetter / extra-token positions against vectors captured from the reference (tests/golden/
make_position_goldens.py; reference: croco/blocks.py:207-219, backbone_masked_croco.py:163-172).  Bit-exact (int64)."""
from pathlib import Path

import torch

from spfsplatv2_amd.rope import PositionGetter, append_token_position

GOLD = torch.load(Path(__file__).parent / "golden" / "position_goldens.pt")


def test_position_getter_matches_reference():
    getter = PositionGetter()
    for case in GOLD.values():
        pos = getter(case["b"], case["h"], case["w"], torch.device("cpu"))
        assert pos.dtype == torch.int64 and pos.is_contiguous()
        assert torch.equal(pos, case["positions"])
    # one private copy per call: writing into a result must not leak into the cache
    a = getter(1, 3, 5, torch.device("cpu"))
    a += 100
    assert torch.equal(getter(2, 3, 5, torch.device("cpu")), GOLD["2x3x5"]["positions"])


def test_extra_token_positions_match_reference():
    for case in GOLD.values():
        one = append_token_position(case["positions"])
        assert torch.equal(one, case["plus_one_token"])
        assert torch.equal(append_token_position(one), case["plus_two_tokens"])
    # the decoder applies the same rule on [b, v, N, 2]
    p4 = GOLD["2x16x16"]["positions"].view(1, 2, 256, 2)
    assert torch.equal(append_token_position(p4)[0], GOLD["2x16x16"]["plus_one_token"])
