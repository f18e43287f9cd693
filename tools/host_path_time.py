#!/usr/bin/env python3
"""The reference's call shape on the realistic scene: TSDF::Integrate(host depth, pose) once per frame (ref:
src/Object.cpp:164) at 512^3, S-surf with a depth frame per pose -- deferred (default) against one kernel per call."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_slam_amd import capi, synth  # noqa: E402

D, vs = 512, 0.005
dims = (D, D, D)
origin = synth.surf_volume(D, vs, 1.0)
scene = synth.SurfScene(dims, vs, origin)
poses = [scene.pose(k, 64) for k in range(64)]
depths = [scene.depth(p, quantize=True) for p in poses]
for defer in (32, 0):
    with capi.Volume(capi.make_config(dims, vs, origin)) as vol:
        vol.set_deferral(defer)
        for k in range(64):
            vol.integrate(depths[k], poses[k])
        vol.sync()
        n = 640
        t0 = time.perf_counter()
        for k in range(n):
            vol.integrate(depths[k % 64], poses[k % 64])
        vol.sync()
        dt = (time.perf_counter() - t0) / n
    print(f"deferral {defer:2d}: {dt * 1e3:.4f} ms per Integrate call, {D ** 3 / dt / 1e6:.0f} Mvox/s")
