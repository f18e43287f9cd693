#!/usr/bin/env python3
"""What a fused launch costs when the classification skips every brick for every frame (the volume is out of view): the
floor under every realistic launch -- dispatch, frame staging, barrier and 32 brick classifications per wavefront."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_slam_amd import capi, synth  # noqa: E402

for D, vs in ((512, 0.005), (200, 0.004), (1024, 0.002)):
    dims = (D, D, D)
    origin = synth.surf_volume(D, vs, 1.0)
    poses = np.stack([synth.make_pose(np.eye(3), [50.0 + 0.01 * k, 0.0, 0.0]) for k in range(32)])   # the volume is far off to the side
    depth = torch.full((480, 640), 2.0, dtype=torch.float32, device="cuda")
    for variant in (8, 7):
        with capi.Volume(capi.make_config(dims, vs, origin)) as vol:
            vol.set_kernel_variant(variant)
            ptrs = [depth.data_ptr()] * 32
            for _ in range(3):
                vol.integrate_frames_device(ptrs, poses)
            vol.sync()
            n = 20
            t0 = time.perf_counter()
            for _ in range(n):
                vol.integrate_frames_device(ptrs, poses)
            vol.sync()
            ms = (time.perf_counter() - t0) / n * 1e3
            w = vol.download()[1]
            assert w.max() == 0 or os.environ.get("TSDF_DEBUG_EXIT")
        print(f"{D}^3, nothing in view, {'classified' if variant == 8 else 'per-voxel  '}: {ms:.3f} ms per 32-frame launch "
              f"({ms / 32:.4f} ms per frame)", flush=True)
