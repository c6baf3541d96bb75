"""development: per-Gaussian comparison of the projection kernel's records with the float64 oracle on one fuzz seed.
    python tools/debug_seed.py 2135 [--plain]"""
import sys
from pathlib import Path
import importlib.util
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from spfsplatv2_amd import rasterizer as rz
from oracle import glue_ref, splat_ref
from tests import util

spec = importlib.util.spec_from_file_location("fuzz_campaign", Path(__file__).resolve().parent / "fuzz_campaign.py")
fc = importlib.util.module_from_spec(spec); spec.loader.exec_module(fc)
seed = int(sys.argv[1])
batch, bg, si, band4, planned, desc = fc.random_case(seed, wide="--plain" not in sys.argv)
print(desc)
b, v = batch.extrinsics.shape[:2]
S, G = batch.opacities.shape
dt = torch.float64
rep = lambda t: t[:, None].expand(b, v, *t.shape[1:]).reshape(b * v, *t.shape[1:])
args = glue_ref.callsite_args(batch.extrinsics.reshape(b * v, 4, 4).to(dt), batch.intrinsics.reshape(b * v, 3, 3).to(dt),
                              batch.near.reshape(-1).to(dt), batch.far.reshape(-1).to(dt), batch.image_shape,
                              torch.tensor(bg)[None].expand(b * v, 3), rep(batch.means).to(dt), rep(batch.harmonics).to(dt),
                              rep(batch.opacities).to(dt), rep(batch.rotations).to(dt), rep(batch.scales).to(dt), scale_invariant=si)
H, W = batch.image_shape
bd = batch.to("cuda")
view, proj, tanfov, vscale = rz.camera_forward(bd.extrinsics, bd.intrinsics, bd.near, bd.far, si)
shs = bd.harmonics.permute(0, 1, 3, 2).contiguous()
K = shs.shape[2]
deg = int(K ** 0.5 + 1e-9) - 1
outs, state, dense = rz._forward_impl(bd.means.contiguous(), bd.scales.contiguous(), bd.rotations.contiguous(), bd.opacities.contiguous(), shs, None,
                                      view, proj, tanfov, rz._background(torch.tensor(bg, device="cuda"), S, v), vscale if si else None,
                                      H, W, deg, 1.0, None, sh_band4=band4)
rec = state[0].cpu().double().reshape(b * v, G, 12)
radii = state[1].cpu().reshape(b * v, G)
worst = {}
for r, a in enumerate(args):
    c = lambda t: None if t is None else t.to(dt)
    pr = splat_ref.project(c(a["means3D"]), c(a["scales"]), c(a["rotations"]), c(a["opacities"]), c(a["shs"]), None, c(a["viewmatrix"]),
                           c(a["projmatrix"]), a["tanfovx"], a["tanfovy"], H, W, a["sh_degree"], 1.0, band4=band4)
    vis = (pr.radii > 0) & (radii[r] > 0)
    con = torch.stack([rec[r, :, 2], rec[r, :, 3], rec[r, :, 4]], -1)
    scale = pr.conic.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    ce = ((con - pr.conic).abs() / scale).amax(dim=1) * vis
    xe = (rec[r, :, :2] - pr.xy).abs().amax(dim=1) * vis
    ze = ((rec[r, :, 6] - pr.depth).abs() / pr.depth.abs().clamp_min(1e-9)) * vis
    ke = (rec[r, :, 8:11] - pr.rgb).abs().amax(dim=1) * vis
    rf = splat_ref.radii_fragile(pr, H, W)
    for gm in torch.nonzero((radii[r] != pr.radii) & ~rf).flatten().tolist():
        print(f"  RADII MISMATCH render {r} g {gm}: oracle {int(pr.radii[gm])} raw {float(pr.radius_raw[gm])!r} xy {pr.xy[gm].tolist()} depth {float(pr.depth[gm])!r} "
              f"rect {pr.rect_min[gm].tolist()} {pr.rect_max[gm].tolist()} | product {int(radii[r, gm])} rec {rec[r, gm, :8].tolist()} scales {a['scales'][gm].tolist()} q {a['rotations'][gm].tolist()}")
    g = int(ce.argmax())
    print(f"render {r}: visible {int(vis.sum())} conic rel err max {float(ce.max()):.3e} (g={g}, radius {int(pr.radii[g])}, conic {pr.conic[g].tolist()} vs {con[g].tolist()}) "
          f"xy {float(xe.max()):.3e} depth rel {float(ze.max()):.3e} rgb {float(ke.max()):.3e}")
ref = util.run_oracle(batch, torch.float64, background=bg, scale_invariant=si, mask_fragile=True, band4=band4)
prod = util.run_product(batch, background=bg, scale_invariant=si, pixel_mask=ref["pixel_mask"], band4=band4)
print(util.compare(prod, ref, max_fragile_frac=0.1))
d = (prod["color"].double() - ref["color"].double()).abs().amax(dim=2) * (~ref["fragile"])
idx = torch.nonzero(d > 1e-4)
print("pixels off (unflagged):", idx.shape[0], "worst", float(d.max()), idx[:10].tolist())
# ---- gradients: who is off? ----
ref32 = util.run_oracle(batch, torch.float32, background=bg, scale_invariant=si, want_fragile=False) if "--f32" in sys.argv else None
gm_ref, gm_prod = ref["grads"]["means"], prod["grads"]["means"].double()
err = (gm_prod - gm_ref).abs().amax(dim=-1)           # [S,G]
print("max |ref grad means|", float(gm_ref.abs().max()), "max err", float(err.max()))
top = torch.topk(err.flatten(), 6).indices
for i in top.tolist():
    s_, g_ = divmod(i, G)
    print(f"scene {s_} gaussian {g_}: err {float(err[s_, g_]):.3e} ref {gm_ref[s_, g_].tolist()} prod {gm_prod[s_, g_].tolist()}",
          "" if ref32 is None else f"oracle-f32 {ref32['grads']['means'][s_, g_].tolist()}")
    for vv in range(v):
        r = s_ * v + vv
        a = args[r]
        c = lambda t: None if t is None else t.to(dt)
        pr = splat_ref.project(c(a["means3D"]), c(a["scales"]), c(a["rotations"]), c(a["opacities"]), c(a["shs"]), None, c(a["viewmatrix"]),
                               c(a["projmatrix"]), a["tanfovx"], a["tanfovy"], H, W, a["sh_degree"], 1.0, band4=band4)
        cov = torch.linalg.inv(torch.tensor([[pr.conic[g_, 0], pr.conic[g_, 1]], [pr.conic[g_, 1], pr.conic[g_, 2]]]))
        ev = torch.linalg.eigvalsh(cov)
        print(f"   view {vv}: radius {int(pr.radii[g_])} xy {pr.xy[g_].tolist()} depth {float(pr.depth[g_]):.4f} opac {float(pr.opacity[g_]):.3f} "
              f"sigma {ev.sqrt().tolist()} scales {a['scales'][g_].tolist()} prod rec {rec[r, g_, :7].tolist()}")
