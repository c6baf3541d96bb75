"""development: time ONLY the forward projection stage (camera + spf_raster_forward_project) -- safe with ablation builds
whose outputs are inconsistent.   SPF_LIB_DIR=_C_p1 python tools/time_project_fwd.py [config] [scenes] [views]"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from spfsplatv2_amd import _lib, rasterizer as R, synthetic as syn

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
V = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
b = syn.make_batch(cfg, S, V, seed=1000).to(dev)
lib = _lib.load()
view, proj, tanfov, vscale = R.camera_forward(b.extrinsics, b.intrinsics, b.near, b.far, True)
G = b.means.shape[1]
K = b.harmonics.shape[-1]
H, W = b.image_shape
deg = int(K ** 0.5) - 1
dims = _lib.SpfDims(S, V, G, K, deg, H, W, 1.0, 1, 0)
T = lib.spf_raster_num_tiles(H, W)
Rr = S * V
i32 = dict(dtype=torch.int32, device=dev)
rec = torch.empty((Rr * G, 12), dtype=torch.float32, device=dev)
radii = torch.empty((Rr * G,), **i32)
rect = torch.empty((2 * Rr * G,), **i32)
nblk = lib.spf_raster_view_partial_blocks(G)
pair_idx = torch.empty((Rr * G + 2 * Rr * nblk,), **i32)
tiles = torch.empty((4 * Rr * T + 1 + 4,), **i32)
bg = torch.zeros(S, V, 3, device=dev)
p = R._ptr
inp = _lib.SpfInputs(p(b.means), p(b.scales), p(b.rotations), p(b.opacities), p(b.harmonics.contiguous()), None, p(view),
                     p(proj), p(tanfov), p(bg), p(vscale))
st = R._state_struct(rec, radii, rect, tiles, None, pair_idx, torch.empty(1, device=dev), torch.empty(1, **i32), Rr * T,
                     Rr * G, Rr * nblk)
stream = R._stream_ptr(dev)
_lib.stage_timing_enable(["project_fwd", "tile_scan"])
for _ in range(30):
    _lib.check(lib.spf_raster_forward_project(C.byref(dims), C.byref(inp), C.byref(st), stream), "project")
torch.cuda.synchronize()
t = _lib.stage_times()
print(" ".join(f"{k}={v[0] / v[1] * 1e3:.1f}" for k, v in t.items() if v[1]), "D", int(tiles[4 * Rr * T + 1]))
