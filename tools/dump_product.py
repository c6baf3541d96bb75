"""development: render one synthetic case with the product and save its outputs for offline comparison with the oracle
    python tools/dump_product.py C5 5 gpurun_out/c5_prod.pt"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from spfsplatv2_amd import synthetic as syn  # noqa: E402
from tests import util  # noqa: E402

cfg, seed, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
batch = syn.make_batch(cfg, 1, 1, seed=seed)
prod = util.run_product(batch, with_grads=False)
torch.save({k: prod[k] for k in ("color", "depth", "alpha", "radii", "stats")}, out)
print(prod["stats"])
