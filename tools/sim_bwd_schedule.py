"""development: CPU model of the lists-backward kernel's schedule on one C2 render (no GPU needed).

Counts, per tile, the replay iterations a wave pays (max over its 64 lanes of the candidates per round) and the
phase-C iterations (max over lanes of the box size), for several lane-assignment / round-size strategies.  Uses the
oracle only as a projector (this is a tool, not product code).

    python tools/sim_bwd_schedule.py [--config C2] [--tiles 64]
"""
from __future__ import annotations

import argparse
import math
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import glue_ref, splat_ref  # noqa: E402
from spfsplatv2_amd import synthetic  # noqa: E402

TILE = 16


def project_render(cfg, seed=0, s_mult=1.0):
    b = synthetic.make_batch(cfg, 1, 1, seed, s_mult)
    rep = lambda t: t.reshape(1, *t.shape[1:])
    bg = torch.zeros(1, 3)
    a = glue_ref.callsite_args(b.extrinsics.reshape(1, 4, 4), b.intrinsics.reshape(1, 3, 3), b.near.reshape(-1),
                               b.far.reshape(-1), b.image_shape, bg, rep(b.means), rep(b.harmonics),
                               rep(b.opacities), rep(b.rotations), rep(b.scales), scale_invariant=True)[0]
    H, W = a["image_height"], a["image_width"]
    with torch.no_grad():
        pr = splat_ref.project(a["means3D"], a["scales"], a["rotations"], a["opacities"], a["shs"], a["colors_precomp"],
                               a["viewmatrix"], a["projmatrix"], a["tanfovx"], a["tanfovy"], H, W, a["sh_degree"],
                               a["scale_modifier"])
    return pr, H, W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--tiles", type=int, default=64)
    ap.add_argument("--s-mult", type=float, default=1.0)
    args = ap.parse_args()
    pr, H, W = project_render(args.config, 0, args.s_mult)
    xy = pr.xy.numpy().astype(np.float64)
    A, B, C = [pr.conic[:, i].numpy().astype(np.float64) for i in range(3)]
    op = pr.opacity.numpy().astype(np.float64)
    depth = pr.depth.numpy()
    vis = pr.radii.numpy() > 0
    # cull disc (project.hip)
    thr = 2.0 * np.log(np.maximum(255.0 * op, 1e-30))
    mu = 0.5 * (A + C) - np.sqrt(0.25 * (A - C) ** 2 + B * B)
    r2 = np.where(255.0 * op > 1.0, np.where(mu > 0, (thr * 1.001 + 1e-3) / np.maximum(mu, 1e-30) * 1.001, 3e38), -1.0)
    rb = np.sqrt(np.maximum(r2, 0)) * 1.0001 + 1e-3
    xlo, xhi = np.ceil(xy[:, 0] - rb), np.floor(xy[:, 0] + rb)
    ylo, yhi = np.ceil(xy[:, 1] - rb), np.floor(xy[:, 1] + rb)
    ok = vis & (r2 >= 0)
    tiles_x, tiles_y = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    rng = np.random.default_rng(0)
    tile_ids = rng.choice(tiles_x * tiles_y, size=min(args.tiles, tiles_x * tiles_y), replace=False)

    stats = {}

    def add(k, v):
        stats[k] = stats.get(k, 0) + v

    for t in tile_ids:
        ty, tx = divmod(int(t), tiles_x)
        X0, Y0 = tx * TILE, ty * TILE
        m = ok & (xhi >= X0) & (xlo <= X0 + TILE - 1) & (yhi >= Y0) & (ylo <= Y0 + TILE - 1)
        # 3-sigma rect too
        m &= (pr.rect_min[:, 0].numpy() <= tx) & (pr.rect_max[:, 0].numpy() > tx) & (pr.rect_min[:, 1].numpy() <= ty) & (pr.rect_max[:, 1].numpy() > ty)
        ids = np.nonzero(m)[0]
        ids = ids[np.argsort(depth[ids], kind="stable")]
        n = len(ids)
        if n == 0:
            continue
        px, py = np.meshgrid(np.arange(X0, X0 + TILE), np.arange(Y0, Y0 + TILE))
        px = px.reshape(-1).astype(np.float64); py = py.reshape(-1).astype(np.float64)        # 256 pixels, row-major
        dx = xy[ids, 0][:, None] - px[None]; dy = xy[ids, 1][:, None] - py[None]
        cand = (dx * dx + dy * dy) <= r2[ids][:, None]                                         # [n,256]
        power = -0.5 * (A[ids][:, None] * dx * dx + C[ids][:, None] * dy * dy) - B[ids][:, None] * dx * dy
        alpha = np.minimum(0.99, op[ids][:, None] * np.exp(power))
        hitc = cand & (power <= 0) & (alpha >= 1 / 255)
        # forward: T, last contributor
        ncon = np.zeros(256, dtype=np.int64); hits = np.zeros(256, dtype=np.int64)
        Tr = np.ones(256)
        done = np.zeros(256, bool)
        for i in range(n):
            h = hitc[i] & ~done
            tt = Tr * (1 - alpha[i])
            stop = h & (tt < 1e-4)
            take = h & ~stop
            done |= stop
            Tr = np.where(take, tt, Tr)
            ncon = np.where(take, i + 1, ncon)
            hits += take
        bmax = int(ncon.max())
        if bmax == 0:
            continue
        # boxes clipped to the tile
        bxl = np.maximum(xlo[ids] - X0, 0); bxh = np.minimum(xhi[ids] - X0, TILE - 1)
        byl = np.maximum(ylo[ids] - Y0, 0); byh = np.minimum(yhi[ids] - Y0, TILE - 1)
        size = (np.maximum(bxh - bxl + 1, 0) * np.maximum(byh - byl + 1, 0)).astype(np.int64)
        candb = cand & (np.arange(n)[:, None] < ncon[None])                                    # replayed candidates
        add("tiles", 1); add("entries", bmax); add("cands", int(candb[:bmax].sum())); add("hits", int(hits.sum()))
        add("slots", int(size[:bmax].sum()))

        order_hits = np.argsort(-hits, kind="stable")                                          # lane assignment (current)

        def rounds(roundL, pool):
            hi = bmax
            out = []
            while hi > 0:
                k = 0; used = 0
                while k < min(roundL, hi) and used + size[hi - 1 - k] <= pool:
                    used += size[hi - 1 - k]; k += 1
                k = max(k, 1)
                out.append((hi - k, hi))
                hi -= k
            return out

        def replay_cost(rs, order_fn, tag, waves=4):
            it = 0; itc = 0; itm = 0
            for (lo, hi_) in rs:
                cc = candb[lo:hi_].sum(0)                                                      # per pixel
                order = order_fn(cc)
                per = 256 // waves
                wmax = 0
                for w in range(waves):
                    lanes = cc[order[w * per:(w + 1) * per]]
                    if per > 64:                                                               # several pixels per lane
                        lanes = lanes.reshape(-1, 64).sum(0) if False else lanes
                    it += int(lanes.max())
                    wmax = max(wmax, int(lanes.max()))
                itm += wmax                                                                    # the round's barrier waits for the slowest wave
                # phase C: entries hi-1-i on thread i: waves of 64 entries
                sz = size[lo:hi_][::-1]
                for w in range(0, len(sz), 64):
                    itc += int(sz[w:w + 64].max())
            add(tag + "_Bslowest", itm)
            add(tag + "_B", it); add(tag + "_C", itc); add(tag + "_rounds", len(rs))

        r_cur = rounds(192, 1536)
        replay_cost(r_cur, lambda cc: order_hits, "cur")
        replay_cost(r_cur, lambda cc: np.arange(256), "natural")
        # static lane orders: wave = one 8x8 pixel block / one 16x4 strip (natural) / 4x16 column strip
        yy, xx = np.divmod(np.arange(256), 16)
        blk8 = np.argsort(((yy // 8) * 2 + (xx // 8)) * 64 + (yy % 8) * 8 + (xx % 8), kind="stable")
        col4 = np.argsort((xx // 4) * 64 + yy * 4 + (xx % 4), kind="stable")
        replay_cost(r_cur, lambda cc: blk8, "block8x8")
        ilv = np.argsort((yy % 4) * 64 + (yy // 4) * 16 + xx, kind="stable")                   # wave w: rows w, w+4, w+8, w+12
        replay_cost(r_cur, lambda cc: ilv, "rows_interleaved")
        replay_cost(r_cur, lambda cc: col4, "col4x16")
        replay_cost(r_cur, lambda cc: np.argsort(-cc, kind="stable"), "perround")
        # forward kernel: natural lane order, rounds of kStage entries front to back (no pool)
        for KS in (256, 384, 512):
            itf = 0
            for lo in range(0, bmax, KS):
                cc = candb[lo:lo + KS].sum(0)
                for w in range(4):
                    itf += int(cc[w * 64:(w + 1) * 64].max())
            add(f"fwd{KS}_B", itf); add(f"fwd{KS}_rounds", (bmax + KS - 1) // KS)
        # forward, balanced rounds: ceil(n / 256) rounds of equal size instead of 256 + 256 + ... + rest
        nr = (bmax + 255) // 256
        per = (bmax + nr - 1) // nr
        itf = 0
        for lo in range(0, bmax, per):
            cc = candb[lo:lo + per].sum(0)
            for w in range(4):
                itf += int(cc[w * 64:(w + 1) * 64].max())
        add("fwd256bal_B", itf)
        r_big = rounds(1 << 30, 1 << 30)
        replay_cost(r_big, lambda cc: order_hits, "whole_hits")
        replay_cost(r_big, lambda cc: np.argsort(-cc, kind="stable"), "whole_cand")
        for RL, PL in ((256, 2048), (384, 3072), (512, 4096)):
            replay_cost(rounds(RL, PL), lambda cc: order_hits, f"r{RL}")
            replay_cost(rounds(RL, PL), lambda cc: np.argsort(-cc, kind="stable"), f"r{RL}pr")

        # two pixels per lane (128-thread replay): pair rank k with rank 255-k
        def pair_cost(rs, tag):
            it = 0
            for (lo, hi_) in rs:
                cc = candb[lo:hi_].sum(0)
                o = order_hits
                lanes = cc[o[:128]] + cc[o[::-1][:128]]
                it += int(lanes[:64].max()) + int(lanes[64:].max())
            add(tag + "_B", it)
        pair_cost(r_cur, "pair")
        pair_cost(r_big, "pair_whole")

    T = stats.pop("tiles")
    print(f"{T} tiles: per tile  entries {stats['entries']/T:.0f}  candidates {stats['cands']/T:.0f}  hits {stats['hits']/T:.0f}  "
          f"slots {stats['slots']/T:.0f}   ideal replay iters/wave {stats['cands']/T/256:.1f}")
    for k in sorted(stats):
        if k.endswith("_Bslowest"):
            print(f"  {k:26s} {stats[k]/T:8.1f}  (slowest wave, summed over the rounds)")
        elif k.endswith("_B") or k.endswith("_C") or k.endswith("_rounds"):
            div = 4 if k.endswith("_B") and not k.startswith("pair") else (2 if k.endswith("_B") else 1)
            print(f"  {k:18s} {stats[k]/T/div:8.1f}" + ("  (per wave)" if k.endswith("_B") else ""))


if __name__ == "__main__":
    main()
