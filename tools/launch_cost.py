#!/usr/bin/env python3
"""Host-side cost of queueing one fused 32-frame launch (poses composed, parameter blocks, tile tables, pre-pass, kernel):
a small volume so that the GPU is never the bottleneck; wall time per call without waiting for the device."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_slam_amd import capi, synth  # noqa: E402

dims, vs = (64, 64, 64), 0.02
origin = synth.surf_volume(64, vs, 1.0)
scene = synth.SurfScene(dims, vs, origin)
poses = np.stack([scene.pose(k, 64) for k in range(32)])
depth = torch.from_numpy(scene.depth(poses[0], quantize=True)).cuda()
ptrs = [depth.data_ptr()] * 32
for variant in (8, 7):
    with capi.Volume(capi.make_config(dims, vs, origin)) as vol:
        vol.set_kernel_variant(variant)
        for _ in range(20):
            vol.integrate_frames_device(ptrs, poses)
        vol.sync()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            vol.integrate_frames_device(ptrs, poses)
        t1 = time.perf_counter()
        vol.sync()
        t2 = time.perf_counter()
    print(f"variant {variant}: {1e6 * (t1 - t0) / n:.1f} us of host time per 32-frame launch queued ({1e6 * (t2 - t0) / n:.1f} us incl. the device)")
