"""Development probe: n object volumes of the reference's default size (200^3 @ 4 mm), one frame --
one batched launch against n per-volume launches.   python tools/batch_time.py [n]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_slam_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
E = int(sys.argv[2]) if len(sys.argv) > 2 else 200      # grid edge of every object volume
use_masks = not (len(sys.argv) > 3 and sys.argv[3] == "nomask")
rng = np.random.default_rng(0)
cfgs = []
for i in range(n):
    o = np.array([-0.4 + rng.uniform(-0.2, 0.2), -0.4 + rng.uniform(-0.2, 0.2), 0.7 + rng.uniform(0, 0.5)], np.float32)
    cfgs.append(capi.make_config((E, E, E), 0.8 / E, o, vol_id=i))
scene = synth.SurfScene((200, 200, 200), 0.004, np.array([-0.4, -0.4, 0.7], np.float32))
depth = torch.from_numpy(scene.depth(scene.pose(0, 8))).cuda()
mask = torch.full((480, 640), 255, dtype=torch.uint8).cuda()
poses = [scene.pose(k, 8) for k in range(8)]
frames = 200
with capi.Batch(cfgs) as batch:
    ptrs = [mask.data_ptr()] * n if use_masks else None
    for k in range(10):
        batch.integrate_device(depth.data_ptr(), ptrs, poses[k % 8])
    batch.sync()
    t0 = time.perf_counter()
    for k in range(frames):
        batch.integrate_device(depth.data_ptr(), ptrs, poses[k % 8])
    batch.sync()
    tb = (time.perf_counter() - t0) / frames
vols = [capi.Volume(c) for c in cfgs]
for k in range(10):
    for v in vols:
        v.integrate_masked_device(depth.data_ptr(), mask.data_ptr(), poses[k % 8]) if use_masks else v.integrate_device(depth.data_ptr(), poses[k % 8])
for v in vols:
    v.sync()
t0 = time.perf_counter()
for k in range(frames):
    for v in vols:
        v.integrate_masked_device(depth.data_ptr(), mask.data_ptr(), poses[k % 8]) if use_masks else v.integrate_device(depth.data_ptr(), poses[k % 8])
for v in vols:
    v.sync()
ts = (time.perf_counter() - t0) / frames
vox = n * E ** 3
print(f"{n} volumes of {E}^3{'' if use_masks else ' (no masks)'}: batched {tb * 1e3:.3f} ms/frame ({vox / tb / 1e6:.0f} Mvox/s), "
      f"per-volume launches {ts * 1e3:.3f} ms/frame ({vox / ts / 1e6:.0f} Mvox/s), speed-up {ts / tb:.2f}x")
