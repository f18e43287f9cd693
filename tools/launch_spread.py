#!/usr/bin/env python3
"""Is the spread of the trajectory's launch times imbalance or work?  Per 32-frame launch of the 194 fr3 keyframes into the
1024^3 volume (second trip along the trajectory): device time, the wavefront-frames that took the per-voxel path, and their
ratio -- a constant ratio says the time follows the work.

    python tools/launch_spread.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from semantic_slam_amd import capi  # noqa: E402

D, vs = 1024, 0.002
dims = (D, D, D)
W = bench.Workload("traj", dims, vs)
n = W.n_pose
dev = [torch.from_numpy(np.ascontiguousarray(d)).cuda() for d in W.depths]
cfg = capi.make_config(dims, vs, W.origin, trunc=W.trunc, base2world=W.base2world)
with capi.Volume(cfg) as vol:
    ptrs = [d.data_ptr() for d in dev]
    vol.integrate_frames_device(ptrs, W.poses)            # first trip: buffers, decisions, a volume that has seen the scene
    vol.sync()
    print(f"traj {D}^3: launch  frames  ms   per-voxel wavefront-frames (M)   free (M)   us per 1000 per-voxel wavefront-frames")
    starts = list(range(0, n, 32))
    counts = []
    for start in starts:                                  # second trip: what each launch does (the counters slow it down 40x)
        m = min(32, n - start)
        vol.shortcut_stats(True)
        vol.integrate_frames_timed(ptrs[start:start + m], W.poses[start:start + m])
        counts.append(vol.shortcut_stats(False))
    rows = []
    for k, start in enumerate(starts):                    # third trip: how long it takes
        m = min(32, n - start)
        ms = vol.integrate_frames_timed(ptrs[start:start + m], W.poses[start:start + m])
        pv, fr, sk = counts[k]
        rows.append((ms, pv))
        print(f"   {k}   {m:3d}   {ms:7.3f}   {pv / 1e6:8.3f}   {fr / 1e6:8.3f}   {ms * 1e3 / max(pv, 1) * 1e3:8.2f}", flush=True)
    ms = np.array([r[0] for r in rows[:-1]]); pv = np.array([r[1] for r in rows[:-1]], np.float64)
    a, b = np.polyfit(pv, ms, 1)
    print(f"full launches: ms = {b:.3f} + {a * 1e6:.3f} per million per-voxel wavefront-frames; correlation {np.corrcoef(pv, ms)[0, 1]:.4f}")
