"""development: render fuzz-campaign cases with the product and save the images for offline calibration of the oracle's
knife-edge flags:    python tools/dump_fuzz_case.py [--wide] out.pt seed [seed ...]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import fuzz_campaign as fc  # noqa: E402
from tests import util  # noqa: E402

args = [a for a in sys.argv[1:] if a != "--wide"]
wide = "--wide" in sys.argv
out = {}
for seed in map(int, args[1:]):
    batch, bg, si, band4, planned, desc = fc.random_case(seed, wide)
    prod = util.run_product(batch, background=bg, scale_invariant=si, with_grads=False, band4=band4)
    out[seed] = {k: prod[k] for k in ("color", "depth", "alpha")}
torch.save(out, args[0])
print("saved", list(out))
