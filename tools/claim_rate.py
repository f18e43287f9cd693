#!/usr/bin/env python3
"""What the patch classification claims on S-surf 512^3 (depth re-rendered per pose), forced on: wavefront-frames that
took the per-voxel path / were updated as free space without projecting a voxel / were skipped -- bricks per wavefront
(variant 8) against rows per workgroup (variant 11)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from semantic_slam_amd import capi, synth  # noqa: E402

D, vs = 512, 0.005
dims = (D, D, D)
origin = synth.surf_volume(D, vs, 1.0)
scene = synth.SurfScene(dims, vs, origin)
poses = np.stack([scene.pose(k, 64) for k in range(64)])
dev = [torch.from_numpy(scene.depth(p, quantize=True)).cuda() for p in poses]
for variant in (8, 11):
    with capi.Volume(capi.make_config(dims, vs, origin)) as vol:
        vol.set_kernel_variant(variant)
        vol.integrate_frames_device([d.data_ptr() for d in dev], poses)      # steady state: second pass counted
        vol.shortcut_stats(True)
        vol.integrate_frames_device([d.data_ptr() for d in dev], poses)
        pv, fr, sk = vol.shortcut_stats(False)
        tot = pv + fr + sk
        print(f"variant {variant}: per-voxel {pv / tot:.3f}  free {fr / tot:.3f}  skipped {sk / tot:.3f}  ({tot} wavefront-frames)")
