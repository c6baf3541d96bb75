"""development: what the rows backward does in a workload (a -DSPF_ROWS_CENSUS build, see tools/experiments/r06_splat_bwd).
   SPF_LIB_DIR=_C_census python tools/rows_census.py --config C2 --s-mult 10"""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
import bench  # noqa: E402
import spfsplatv2_amd as spf  # noqa: E402
from spfsplatv2_amd import _lib, synthetic as syn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2")
ap.add_argument("--s-mult", type=float, default=10.0)
a = ap.parse_args()
S, V = bench.WORKLOADS[a.config]
b = syn.make_batch(a.config, S, V, seed=1000, s_mult=a.s_mult).to("cuda")
lib = C.CDLL(str(_lib.LIB_PATH))
out = (C.c_ulonglong * 8)()
dec = spf.DecoderSplattingCUDA(spf.DecoderSplattingCUDACfg(name="splatting_cuda", background_color=[0.0, 0.0, 0.0], make_scale_invariant=False, enable_cov_grad=False, enable_sh_grad=False)).to("cuda")
g = spf.Gaussians(b.means.requires_grad_(True), None, b.rotations, b.scales, b.harmonics, b.opacities)
o, alpha, _ = dec.render(g, b.extrinsics, b.intrinsics, b.near, b.far, b.image_shape)
torch.cuda.synchronize()
lib.spf_debug_rows_census(out, 1)
(o.color.square().sum() + o.depth.sum()).backward()
torch.cuda.synchronize()
lib.spf_debug_rows_census(out, 1)
tiles, bmax, steps, rowsteps, hits, heavy, qsteps, n = [int(x) for x in out]
print(json.dumps(dict(config=a.config, s_mult=a.s_mult, dense_tiles=tiles, mean_bmax=bmax / max(tiles, 1), mean_n=n / max(tiles, 1),
                      wave_steps=steps, wave_steps_with_a_hit=heavy, row_steps=rowsteps, hits=hits,
                      unculled_wave_steps=4 * bmax, quad_walk_wave_steps=qsteps, quad_over_rows=qsteps / max(steps, 1), hit_fraction_of_all_pairs=hits / max(256 * bmax, 1),
                      rows_steps_over_unculled=steps / max(4 * bmax, 1))))
