"""Development probe: n object volumes of the reference's default size (200^3 @ 4 mm), one frame per call -- the
reference's real call shape (one TSDF per object instance fed depth x its instance mask, ref: src/Engine.cpp:172-233,
src/Object.cpp:67) -- as one batched launch and as n per-volume launches, each with and without the per-workgroup
classification of masked frames (kernel variant 0: the library's size policy; 8: always; 11: by policy with round 2a's
1024-voxel workgroup patches instead of wavefront bricks; 7: never).

    python tools/batch_time.py [--n 16] [--edge 200] [--masks instance|full|none]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_slam_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16)
ap.add_argument("--edge", type=int, default=200)
ap.add_argument("--masks", default="instance", choices=["instance", "full", "none"])
ap.add_argument("--frames", type=int, default=200)
args = ap.parse_args()
n, E = args.n, args.edge
rng = np.random.default_rng(0)
vs = 0.8 / E
cfgs, masks = [], []
K = synth.TUM_K
for i in range(n):
    o = np.array([-0.4 + rng.uniform(-0.3, 0.3), -0.4 + rng.uniform(-0.25, 0.25), 0.7 + rng.uniform(0, 0.8)], np.float32)
    cfgs.append(capi.make_config((E, E, E), vs, o, vol_id=i))
    m = np.zeros((480, 640), np.uint8)
    if args.masks == "full":
        m[:] = 255
    elif args.masks == "instance":
        # the instance: a 0.5 m box around the volume's centre as the first camera sees it
        c = o + 0.4
        u0, u1 = K[0] * (c[0] - 0.25) / c[2] + K[2], K[0] * (c[0] + 0.25) / c[2] + K[2]
        v0, v1 = K[4] * (c[1] - 0.25) / c[2] + K[5], K[4] * (c[1] + 0.25) / c[2] + K[5]
        m[max(0, int(v0)):max(0, min(480, int(v1))), max(0, int(u0)):max(0, min(640, int(u1)))] = 255
    masks.append(m)
scene = synth.SurfScene((200, 200, 200), 0.004, np.array([-0.4, -0.4, 0.7], np.float32))
depth = torch.from_numpy(scene.depth(scene.pose(0, 8))).cuda()
m_dev = [torch.from_numpy(m).cuda() for m in masks]
ptrs = None if args.masks == "none" else [m.data_ptr() for m in m_dev]
poses = [scene.pose(k, 8) for k in range(8)]
frames = args.frames
cover = float(np.mean([m.mean() / 255.0 for m in masks]))
vox = n * E ** 3
print(f"{n} volumes of {E}^3, masks: {args.masks} (mean coverage {cover:.2f} of the image)")
NAMES = {0: "by policy (bricks)        ", 8: "forced on (bricks)        ", 11: "by policy (1024-voxel patches)", 7: "off                       ", 13: "round-2 brick workgroups   "}
for cls, deferral in [(0, 32)] + ([(13, 32)] if capi.experiments_build() else []) + [(0, 0), (8, 0)] + ([(11, 0)] if capi.experiments_build() else []) + [(7, 0)]:
    with capi.Batch(cfgs) as batch:
        for v in batch.volumes:
            v.set_kernel_variant(cls)
        batch.volumes[0].set_deferral(deferral)   # 32: few / large members collect frames and fuse 32 per launch (the default)
        for k in range(10):
            batch.integrate_device(depth.data_ptr(), ptrs, poses[k % 8])
        batch.sync()
        t0 = time.perf_counter()
        for k in range(frames):
            batch.integrate_device(depth.data_ptr(), ptrs, poses[k % 8])
        batch.sync()
        tb = (time.perf_counter() - t0) / frames
        upd = sum(float(v.download()[1].sum()) for v in batch.volumes) / (frames + 10)
    how = "batch, deferral by policy        " if deferral else "batched launch per frame         "
    print(f"  {how}, classification {NAMES[cls]}: {tb * 1e3:.3f} ms/frame ({vox / tb / 1e6:.0f} Mvox/s), "
          f"{upd / vox:.3f} of the voxels updated per frame", flush=True)
for cls in (0, 8, 7):
    vols = [capi.Volume(c) for c in cfgs]
    for v in vols:
        v.set_kernel_variant(cls)

    def one(k):
        for i, v in enumerate(vols):
            if ptrs is None:
                v.integrate_device(depth.data_ptr(), poses[k % 8])
            else:
                v.integrate_masked_device(depth.data_ptr(), ptrs[i], poses[k % 8])
    for k in range(10):
        one(k)
    for v in vols:
        v.sync()
    t0 = time.perf_counter()
    for k in range(frames):
        one(k)
    for v in vols:
        v.sync()
    ts = (time.perf_counter() - t0) / frames
    for v in vols:
        v.close()
    print(f"  per-volume launches, classification {NAMES[cls]}: {ts * 1e3:.3f} ms/frame ({vox / ts / 1e6:.0f} Mvox/s)", flush=True)
