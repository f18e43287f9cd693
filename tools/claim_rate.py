#!/usr/bin/env python3
"""What the brick classification claims, forced on (variant 8), by brick shape: wavefront-frames that took the per-voxel
path / were updated as free space without projecting a voxel / were skipped, and the time per frame of the fused path.

    python tools/claim_rate.py [--workload ssurf|traj] [--grid 512] [--shapes 8,8,1:4,8,2:2,4,8]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from semantic_slam_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="ssurf")
ap.add_argument("--grid", type=int, default=512)
ap.add_argument("--shapes", default="16,4,1:8,8,1:4,8,2:4,4,4:2,4,8")
ap.add_argument("--noise-mm", type=float, default=0.0)
ap.add_argument("--holes", type=float, default=0.0)
a = ap.parse_args()
D = a.grid
vs = {512: 0.005, 1024: 0.002}.get(D, 2.56 / D)
dims = (D, D, D)
W = bench.Workload(a.workload, dims, vs, a.noise_mm, a.holes)
n = min(64, W.n_pose)
poses = W.poses[:n]
dev = [torch.from_numpy(np.ascontiguousarray(W.depths[i % len(W.depths)])).cuda() for i in range(n)]
cfg = capi.make_config(dims, vs, W.origin, trunc=W.trunc, base2world=W.base2world)
print(f"{a.workload} {D}^3 @ {vs * 1000:g} mm, {n} frames" + (f", noise {a.noise_mm:g} mm, holes {a.holes:g}" if a.noise_mm or a.holes else ""))
for shape in a.shapes.split(":"):
    q, r, s = (int(x) for x in shape.split(","))
    with capi.Volume(cfg) as vol:
        vol.set_kernel_variant(8)
        vol.set_brick_shape(q, r, s)
        ptrs = [d.data_ptr() for d in dev]
        vol.integrate_frames_device(ptrs, poses)      # steady state: second pass counted
        vol.shortcut_stats(True)
        vol.integrate_frames_device(ptrs, poses)
        bl = vol.brick_list_stats()
        pv, fr, sk = vol.shortcut_stats(False)
        tot = pv + fr + sk
        vol.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            vol.integrate_frames_device(ptrs, poses)
        vol.sync()
        ms = (time.perf_counter() - t0) / (3 * n) * 1e3
        print(f"  brick {shape:8s}: per-voxel {pv / tot:.4f}  free {fr / tot:.4f}  skipped {sk / tot:.4f}  "
              f"({tot} wavefront-frames)  {ms:.4f} ms/frame\n"
              f"      work list over {n // 32} launches: super-bricks every frame skipped {bl[0]}, bricks listed {bl[1]}, of those left to "
              f"classify themselves {bl[2]}, of those skipped after all {bl[3]}", flush=True)
