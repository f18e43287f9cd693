#!/usr/bin/env python3
"""hipGraph or not for launch-bound sequences: 32 one-frame launches on small volumes, queued call by call against
replayed from a captured graph (tsdf_probe_graph_replay).  Device time per frame by HIP events."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from semantic_slam_amd import capi, synth  # noqa: E402

for E in (64, 100, 200, 256):
    vs = 0.8 / E
    dims = (E, E, E)
    origin = synth.surf_volume(E, vs, 0.7)
    scene = synth.SurfScene(dims, vs, origin)
    poses = np.stack([scene.pose(k, 32) for k in range(32)])
    depth = torch.from_numpy(scene.depth(poses[0], quantize=True)).cuda()
    with capi.Volume(capi.make_config(dims, vs, origin)) as vol:
        vol.set_kernel_variant(3)           # one kernel per frame
        a, b = vol.probe_graph_replay(depth.data_ptr(), poses, iters=20)
        print(f"{E}^3: 32 launches queued call by call {a / 32 * 1e3:.2f} us per frame, replayed from a hipGraph {b / 32 * 1e3:.2f} us per frame")
