"""The HIP rasterizer against the oracle's FROZEN values (tests/golden/oracle_known_answers.pt): BASELINE config 1 and
three small scenes, forward and all six gradients -- no live oracle run is involved, so this comparison cannot move
with an oracle edit (tests/test_oracle_known_answers.py keeps the live oracle on the same values).
Tolerances: north_star (RGB 1e-4 absolute, gradients 1e-3 of the tensor's scale), radii bit-exact off knife edges."""
import pytest
import torch

from tests import util
from tests.test_oracle_known_answers import CASES, load_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", CASES)
def test_product_matches_frozen_oracle_values(hip_lib, golden_dir, name):
    case, batch = load_case(golden_dir, name)
    want = dict(case["expect"])
    mask = (~want["fragile"]).to(torch.float32)
    prod = util.run_product(batch, background=case["background"], scale_invariant=case["scale_invariant"],
                            pixel_mask=mask, band4=case["band4"])
    rep = util.compare(prod, want, max_fragile_frac=0.005)
    assert not rep["fails"], rep
    assert abs(prod["loss"] - want["loss"]) < 2e-5 * abs(want["loss"]), (prod["loss"], want["loss"])
