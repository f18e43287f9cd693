#!/usr/bin/env python3
"""claim_sim.py -- CPU study (numpy, no GPU): what share of a frame's wavefront bricks could a box classifier decide?

For one S-surf frame and grid the per-voxel outcome of Integrate is computed (none / updated with dist = 1 / inside the
truncation band), then the brick-level claim rule of classify_patch (csrc/tsdf_multiframe.hip.h) is replayed with depth tiles of
8, 4 and 1 pixels, for the brick as a whole and as the conjunction of per-lane claims (one 4 x 1 x 1 quad per lane: a brick is
settled when every lane is).  Statistics only -- margins as the library derives them, arithmetic in float64 -- used to decide
which refinement of the classification is worth building.

    python tools/claim_sim.py [--grid 512] [--frame 10] [--shape 8,4,8]
"""
import argparse
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_slam_amd import synth  # noqa: E402


def sparse_tables(tmin, tmax):
    """2-D sparse tables (dict[(ky, kx)] -> (min table, max table)) over tile arrays."""
    th, tw = tmin.shape
    out = {(0, 0): (tmin, tmax)}
    ly, lx = int(math.floor(math.log2(th))) + 1, int(math.floor(math.log2(tw))) + 1
    for kx in range(1, lx):
        a, b = out[(0, kx - 1)]
        s = 1 << (kx - 1)
        idx = np.minimum(np.arange(tw) + s, tw - 1)
        out[(0, kx)] = (np.minimum(a, a[:, idx]), np.maximum(b, b[:, idx]))
    for ky in range(1, ly):
        s = 1 << (ky - 1)
        idy = np.minimum(np.arange(th) + s, th - 1)
        for kx in range(lx):
            a, b = out[(ky - 1, kx)]
            out[(ky, kx)] = (np.minimum(a, a[idy, :]), np.maximum(b, b[idy, :]))
    return out, ly, lx


class Tiles:
    def __init__(self, depth, T, max_depth=6.0):
        H, W = depth.shape
        th, tw = (H + T - 1) // T, (W + T - 1) // T
        pad = np.full((th * T, tw * T), np.nan, np.float64)
        pad[:H, :W] = depth
        blk = pad.reshape(th, T, tw, T).transpose(0, 2, 1, 3).reshape(th, tw, T * T)
        inimg = ~np.isnan(blk)
        valid = inimg & (blk > 0) & (blk <= max_depth)
        allvalid = np.all(valid | ~inimg, axis=2)
        mn = np.where(valid, blk, np.inf).min(axis=2)
        mx = np.where(valid, blk, -np.inf).max(axis=2)
        self.T, self.th, self.tw = T, th, tw
        tmin = np.where(allvalid, mn, -np.inf)
        # stack the levels into one array for fancy indexing
        tabs, self.ly, self.lx = sparse_tables(tmin, mx)
        self.mn = np.stack([np.stack([tabs[(ky, kx)][0] for kx in range(self.lx)]) for ky in range(self.ly)])
        self.mx = np.stack([np.stack([tabs[(ky, kx)][1] for kx in range(self.lx)]) for ky in range(self.ly)])

    def query(self, u0, u1, v0, v1, W, H):
        """dmin (-inf unless every pixel valid), dmax over the pixel rectangle (already clipped to the image)."""
        T = self.T
        tx0, tx1 = (np.clip(u0, 0, W - 1) // T).astype(np.int64), (np.clip(u1, 0, W - 1) // T).astype(np.int64)
        ty0, ty1 = (np.clip(v0, 0, H - 1) // T).astype(np.int64), (np.clip(v1, 0, H - 1) // T).astype(np.int64)
        tx1, ty1 = np.maximum(tx1, tx0), np.maximum(ty1, ty0)
        kx = np.floor(np.log2(tx1 - tx0 + 1)).astype(np.int64)
        ky = np.floor(np.log2(ty1 - ty0 + 1)).astype(np.int64)
        xb, yb = tx1 - (1 << kx) + 1, ty1 - (1 << ky) + 1
        dmin = np.minimum(np.minimum(self.mn[ky, kx, ty0, tx0], self.mn[ky, kx, ty0, xb]),
                          np.minimum(self.mn[ky, kx, yb, tx0], self.mn[ky, kx, yb, xb]))
        dmax = np.maximum(np.maximum(self.mx[ky, kx, ty0, tx0], self.mx[ky, kx, ty0, xb]),
                          np.maximum(self.mx[ky, kx, yb, tx0], self.mx[ky, kx, yb, xb]))
        return dmin, dmax


def classify(tiles, umin, umax, vmin, vmax, czmin, czmax, mu, mv, W, H, trunc, pad, cz_short):
    """classify_patch's rule on arrays of boxes: 1 free, 2 skip, 0 undecided."""
    u0, u1, v0, v1 = umin - mu, umax + mu, vmin - mv, vmax + mv
    ok = czmin > cz_short
    miss = ~(u1 >= 0) | ~(v1 >= 0) | ~(u0 <= W - 1) | ~(v0 <= H - 1)
    inside = (u0 >= 0) & (v0 >= 0) & (u1 <= W - 1) & (v1 <= H - 1)
    dmin, dmax = tiles.query(np.nan_to_num(u0, nan=0.0), np.nan_to_num(u1, nan=0.0), np.nan_to_num(v0, nan=0.0), np.nan_to_num(v1, nan=0.0), W, H)
    free = inside & (dmin >= czmax + pad + trunc)
    skip = dmax <= czmin - pad - trunc
    out = np.zeros(umin.shape, np.int8)
    out[free] = 1
    out[skip & ~free] = 2
    out[miss] = 2
    # 3 = "free wherever inside": the box straddles an image border, every pixel of its clipped pixel box is valid and at
    # least trunc deeper than the farthest corner -- each voxel is updated with dist = 1 iff its pixel lies inside the image
    out[(out == 0) & ~inside & (dmin >= czmax + pad + trunc)] = 3
    out[~ok] = 0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--frame", type=int, default=10)
    ap.add_argument("--shape", default="8,4,8", help="brick: voxels along x, y, z (x a multiple of 4; 256 voxels)")
    ap.add_argument("--tiles", default="8,4,1")
    ap.add_argument("--noise-mm", type=float, default=0.0)
    ap.add_argument("--holes", type=float, default=0.0)
    a = ap.parse_args()
    D = a.grid
    vs = {512: 0.005, 1024: 0.002}.get(D, 2.56 / D)
    bx, by, bz = (int(x) for x in a.shape.split(","))
    assert bx * by * bz == 256 and bx % 4 == 0
    dims = (D, D, D)
    origin = synth.surf_volume(D, vs, 1.0)
    scene = synth.SurfScene(dims, vs, origin)
    pose = scene.pose(a.frame, 64)
    depth = scene.depth(pose, quantize=True)
    if a.noise_mm > 0 or a.holes > 0:
        depth = synth.sensor_imperfections([depth], a.noise_mm, a.holes)[0]
    depth = depth.astype(np.float64)
    H, W = depth.shape
    K = synth.TUM_K.astype(np.float64)
    fx, fy, cx, cy = K[0], K[4], K[2], K[5]
    T4 = pose.reshape(4, 4).astype(np.float64)
    R, t = T4[:3, :3], T4[:3, 3]
    trunc = 5 * vs
    mu = 0.5625 + 3.2e-5 * (abs(fx) + 4.0 * (W + abs(cx)))
    mv = 0.5625 + 3.2e-5 * (abs(fy) + 4.0 * (H + abs(cy)))
    pad, cz_short = 1e-5, 0.08
    tile_sizes = [int(x) for x in a.tiles.split(",")]
    tiles = {T: Tiles(depth, T) for T in tile_sizes}

    xs = origin[0] + np.arange(D) * vs - t[0]
    ys = origin[1] + np.arange(D) * vs - t[1]
    tot = dict(bricks=0, band=0, none=0, free=0, mixed=0)
    und = {("brick", T): 0 for T in tile_sizes}
    und.update({("lane", T): 0 for T in tile_sizes})
    und_noband = dict(und)
    border = {T: 0 for T in tile_sizes}              # bricks a "free wherever inside the image" claim would settle
    lane_und_count = {T: 0 for T in tile_sizes}     # undecided lanes in bricks the lane-level test leaves undecided
    for z0 in range(0, D, bz):
        zs = origin[2] + (z0 + np.arange(bz)) * vs - t[2]
        dz, dy, dx = np.meshgrid(zs, ys, xs, indexing="ij")
        pcx = R[0, 0] * dx + R[1, 0] * dy + R[2, 0] * dz
        pcy = R[0, 1] * dx + R[1, 1] * dy + R[2, 1] * dz
        pcz = R[0, 2] * dx + R[1, 2] * dy + R[2, 2] * dz
        del dx, dy, dz
        with np.errstate(divide="ignore", invalid="ignore"):
            u = fx * (pcx / pcz) + cx
            v = fy * (pcy / pcz) + cy
        del pcx, pcy
        iu, iv = np.floor(u + 0.5), np.floor(v + 0.5)
        ok = (pcz > 0) & (iu >= 0) & (iu < W) & (iv >= 0) & (iv < H)
        d = np.zeros_like(u)
        d[ok] = depth[iv[ok].astype(np.int64), iu[ok].astype(np.int64)]
        diff = d - pcz
        upd = ok & (d > 0) & (d <= 6.0) & (diff > -trunc)
        band = upd & (diff < trunc)
        del iu, iv, d, diff, ok

        def boxes(arr, shape, fn):
            sx, sy, sz = shape
            r = arr.reshape(bz // sz, sz, D // sy, sy, D // sx, sx)
            return fn(fn(fn(r, axis=5), axis=3), axis=1)

        shp = (bx, by, bz)
        b_band = boxes(band, shp, np.any)
        b_any = boxes(upd, shp, np.any)
        b_all = boxes(upd, shp, np.all)
        nb = b_band.size
        tot["bricks"] += nb
        tot["band"] += int(b_band.sum())
        tot["none"] += int((~b_any).sum())
        tot["free"] += int((b_all & ~b_band).sum())
        tot["mixed"] += int((b_any & ~b_all & ~b_band).sum())
        bb = [boxes(u, shp, np.min), boxes(u, shp, np.max), boxes(v, shp, np.min), boxes(v, shp, np.max),
              boxes(pcz, shp, np.min), boxes(pcz, shp, np.max)]
        lshp = (4, 1, 1)
        lb = [boxes(u, lshp, np.min), boxes(u, lshp, np.max), boxes(v, lshp, np.min), boxes(v, lshp, np.max),
              boxes(pcz, lshp, np.min), boxes(pcz, lshp, np.max)]
        for T in tile_sizes:
            cb = classify(tiles[T], *bb, mu, mv, W, H, trunc, pad, cz_short)
            border[T] += int((cb == 3).sum())
            ub = (cb == 0) | (cb == 3)
            und[("brick", T)] += int(ub.sum())
            und_noband[("brick", T)] += int((ub & ~b_band).sum())
            cl = classify(tiles[T], *lb, mu, mv, W, H, trunc, pad, cz_short)        # [bz, D, D/4]
            lane_und = ((cl == 0) | (cl == 3)).reshape(bz // bz, bz, D // by, by, D // bx, bx // 4)
            n_und = lane_und.sum(axis=(1, 3, 5))
            ul = (n_und > 0) & ub                 # the lane-level test runs on the bricks their own box left undecided
            und[("lane", T)] += int(ul.sum())
            und_noband[("lane", T)] += int((ul & ~b_band).sum())
            lane_und_count[T] += int(n_und[ul].sum())
        print(f"z {z0:4d}: bricks {tot['bricks']}  band {tot['band']}  undecided brick/8 {und.get(('brick', 8), 0)}", flush=True)
    n = tot["bricks"]
    print(f"\nS-surf {D}^3 @ {vs * 1000:g} mm, frame {a.frame}, brick {bx}x{by}x{bz}, noise {a.noise_mm} mm, holes {a.holes}")
    print(f"bricks {n}: with a band voxel {tot['band']} ({100 * tot['band'] / n:.2f} %), nothing updated {100 * tot['none'] / n:.2f} %, "
          f"all dist=1 {100 * tot['free'] / n:.2f} %, mixed without band {100 * tot['mixed'] / n:.2f} %")
    for T in tile_sizes:
        print(f"  of the undecided bricks, {T}-pixel tiles: {border[T]} ({100 * border[T] / n:.2f} % of the bricks) straddle an image border over "
              "valid far pixels only (\"free wherever inside the image\")")
    for key in und:
        lvl, T = key
        extra = f", {lane_und_count[T] / max(und[key], 1):.1f} undecided lanes per such brick" if lvl == "lane" else ""
        print(f"  {lvl:5s} claims, {T}-pixel tiles: undecided {und[key]} ({100 * und[key] / n:.2f} % of the bricks), "
              f"of them without a band voxel {und_noband[key]} ({100 * und_noband[key] / n:.2f} %){extra}")


if __name__ == "__main__":
    main()
